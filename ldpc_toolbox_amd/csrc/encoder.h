// Systematic LDPC encoder (host side of the frame pipeline feeding the decode path).
//
// Behaviour follows /root/reference/src/encoder.rs:59-120: H = [H0 H1] with H1 the
// last (n-k) columns; staircase codes (encoder/staircase.rs:3-24) get the O(n)
// accumulate encoder, every other code gets the dense generator G0 = H1^-1 H0 from a
// GF(2) Gauss reduction (linalg.rs:8-66) -- here bit-packed in 64-bit words instead
// of the reference's byte-per-element ndarray, which is what makes 5G NR BG1 Zc=384
// (17664 x 26112) tractable.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "sparse.h"

namespace ldpc {

bool is_staircase(const SparseMatrix &h);

class Encoder {
 public:
  // Returns false (and sets *err) when the last n-k columns are not invertible.
  static bool from_h(const SparseMatrix &h, Encoder *out, std::string *err);

  size_t k() const { return k_; }
  size_t n() const { return n_; }
  bool staircase() const { return staircase_; }

  // message: k bytes (0/1); codeword: n bytes (0/1) = message followed by parity.
  void encode(const uint8_t *message, uint8_t *codeword) const;

 private:
  size_t k_ = 0, n_ = 0;
  bool staircase_ = false;
  // staircase: rows of H0 as index lists
  std::vector<uint32_t> h0_ptr_, h0_idx_;
  // dense: (n-k) rows of ceil(k/64) words
  size_t words_ = 0;
  std::vector<uint64_t> gen_;
};

}  // namespace ldpc
