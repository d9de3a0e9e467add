#include "encoder.h"

#include <algorithm>

namespace ldpc {

bool is_staircase(const SparseMatrix &h) {
  const size_t m = h.num_rows(), n = h.num_cols();
  if (m > n) return false;
  const size_t k = n - m;
  size_t ones_in_parity = 0;
  for (size_t r = 0; r < m; r++) {
    for (uint32_t c : h.row(r)) {
      if (c < k) continue;
      if (r == 0 && c != k) return false;
      if (r != 0 && c != k + r - 1 && c != k + r) return false;
      ones_in_parity++;
    }
  }
  return m > 0 && ones_in_parity == 2 * m - 1;
}

bool Encoder::from_h(const SparseMatrix &h, Encoder *out, std::string *err) {
  const size_t m = h.num_rows(), n = h.num_cols();
  if (m > n) {
    if (err) *err = "parity check matrix has more rows than columns";
    return false;
  }
  Encoder enc;
  enc.n_ = n;
  enc.k_ = n - m;
  const size_t k = enc.k_;
  if (is_staircase(h)) {
    enc.staircase_ = true;
    enc.h0_ptr_.resize(m + 1);
    for (size_t r = 0; r < m; r++) {
      enc.h0_ptr_[r] = static_cast<uint32_t>(enc.h0_idx_.size());
      for (uint32_t c : h.row(r))
        if (c < k) enc.h0_idx_.push_back(c);
    }
    enc.h0_ptr_[m] = static_cast<uint32_t>(enc.h0_idx_.size());
    *out = std::move(enc);
    return true;
  }
  // A = [H1 H0], m x n bits, reduced to [I G0]
  const size_t words = (n + 63) / 64;
  std::vector<uint64_t> a(m * words, 0);
  for (size_t r = 0; r < m; r++)
    for (uint32_t c : h.row(r)) {
      const size_t t = c < k ? c + m : c - k;
      a[r * words + (t >> 6)] ^= uint64_t{1} << (t & 63);
    }
  auto bit = [&](size_t r, size_t c) { return (a[r * words + (c >> 6)] >> (c & 63)) & 1; };
  for (size_t j = 0; j < m; j++) {
    size_t p = j;
    while (p < m && !bit(p, j)) p++;
    if (p == m) {
      if (err)
        *err = "the square matrix formed by the last columns of the parity check is not invertible";
      return false;
    }
    const size_t w0 = j >> 6;
    if (p != j)
      for (size_t w = w0; w < words; w++) std::swap(a[j * words + w], a[p * words + w]);
    const uint64_t *src = &a[j * words];
    const uint64_t mask = uint64_t{1} << (j & 63);
    for (size_t t = 0; t < m; t++) {
      if (t == j) continue;
      uint64_t *dst = &a[t * words];
      if (!(dst[w0] & mask)) continue;
      for (size_t w = w0; w < words; w++) dst[w] ^= src[w];
    }
  }
  // G0 = columns m..n of the reduced matrix, re-packed from bit 0
  enc.words_ = (k + 63) / 64;
  enc.gen_.assign(m * enc.words_, 0);
  for (size_t r = 0; r < m; r++)
    for (size_t c = 0; c < k; c++)
      if (bit(r, m + c)) enc.gen_[r * enc.words_ + (c >> 6)] |= uint64_t{1} << (c & 63);
  *out = std::move(enc);
  return true;
}

void Encoder::encode(const uint8_t *message, uint8_t *codeword) const {
  const size_t m = n_ - k_;
  std::copy(message, message + k_, codeword);
  uint8_t *parity = codeword + k_;
  if (staircase_) {
    uint8_t acc = 0;
    for (size_t r = 0; r < m; r++) {
      uint8_t s = 0;
      for (uint32_t i = h0_ptr_[r]; i < h0_ptr_[r + 1]; i++) s ^= message[h0_idx_[i]] & 1;
      acc ^= s;
      parity[r] = acc;
    }
    return;
  }
  std::vector<uint64_t> packed(words_, 0);
  for (size_t c = 0; c < k_; c++)
    if (message[c] & 1) packed[c >> 6] |= uint64_t{1} << (c & 63);
  for (size_t r = 0; r < m; r++) {
    const uint64_t *g = &gen_[r * words_];
    uint64_t x = 0;
    for (size_t w = 0; w < words_; w++) x ^= g[w] & packed[w];
    parity[r] = static_cast<uint8_t>(__builtin_popcountll(x) & 1);
  }
}

}  // namespace ldpc
