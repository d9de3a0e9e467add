// GPU-resident BER simulation step: generate frames -> decode -> count errors, all on the device.
// Host-side counterpart of the reference's BerTest / Worker (src/simulation/ber.rs:246-282, 297-368,
// 436-481) for BPSK and 8PSK (with the DVB-S2 bit interleaver) over AWGN; see frame_gen.hip.h for
// the frame definition.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "device_decoder.h"
#include "encoder.h"

namespace ldpc {

class Simulator {
 public:
  // pool: number of pre-encoded random messages frames draw their codeword from (the decoders are
  // symmetric, so the error statistics do not depend on which codewords are sent)
  static Simulator *create(const std::string &alist, const std::string &implementation,
                           const std::string &puncturing, int device, uint32_t pool, uint64_t pool_seed,
                           std::string *err);
  ~Simulator();

  size_t k() const { return k_; }
  size_t n() const { return n_; }
  size_t n_tx() const { return n_tx_; }
  uint32_t pool() const { return pool_; }
  double rate() const { return static_cast<double>(k_) / static_cast<double>(n_tx_); }  // ber.rs:259
  // modulation: 1 = BPSK, 3 = 8PSK (bits per symbol, factory.rs:53-73; 8PSK needs n_tx % 3 == 0).
  // interleaving: columns of the DVB-S2 bit interleaver, negative = rows read backwards, 0 = none
  // (ber.rs:250-252; needs n_tx % columns == 0).  Both return false on an unusable value.
  bool set_modulation(int bits_per_symbol);
  bool set_interleaving(int64_t columns);
  int modulation() const { return bits_per_symbol_; }
  int64_t interleaving() const { return interleaving_; }
  DeviceDecoder *decoder() { return dec_.get(); }
  // frames the last run() / run_bch() call put through the continuous-batching path (0: drained batches)
  uint64_t streamed_frames() const { return streamed_frames_; }
  void set_streaming(bool on) { streaming_ = on; }
  // straggler pooling (run_bch): 0 = every chunk runs the full iteration budget
  void set_pooling(bool on) {
    pooling_ = on;
    budget_valid_ = false;
  }
  uint64_t pooled_frames() const { return pooled_frames_; }
  const std::vector<uint8_t> &messages() const { return messages_; }
  const std::vector<uint8_t> &tx_bits() const { return tx_bits_; }
  const std::string &last_error() const { return error_; }

  // Frames [first_frame, first_frame + frames) at ebn0_db: returns the six counters of
  // sharding.COUNTER_FIELDS (frames, bit errors, frame errors, false decodes, total iterations,
  // iterations of the correct frames) for exactly these frames.
  int run(double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames, uint32_t max_iterations,
          uint64_t counters[6]);
  // The same with the outer-BCH accounting of ber.rs:328-337: counters[9] = the six above, then
  // BCH bit errors, BCH frame errors (frames with more than bch_max_errors bit errors) and the
  // iterations of the frames the BCH code corrects.
  int run_bch(double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames, uint32_t max_iterations,
              uint64_t bch_max_errors, uint64_t counters[9]);
  // The same frames' LLRs (host [frames][n_tx]) and pooled-codeword indices, for tests.
  int generate(double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames, float *llrs,
               uint32_t *pool_index);

 private:
  Simulator() = default;
  int ensure(size_t frames, size_t llr_rows);
  void noise_params(double ebn0_db, float *sigma, float *scale) const;
  double noise_sigma(double ebn0_db) const;
  void launch_generator(double ebn0_db, uint64_t seed, uint64_t first_frame, uint32_t frames, float *dst = nullptr);
  bool fail(const std::string &m, hipError_t e = hipSuccess);

  std::unique_ptr<DeviceDecoder> dec_;
  size_t k_ = 0, n_ = 0, n_tx_ = 0;
  uint32_t pool_ = 0;
  int device_ = 0;
  int bits_per_symbol_ = 1;
  int64_t interleaving_ = 0;
  std::vector<uint8_t> messages_, tx_bits_;
  uint8_t *d_messages_ = nullptr, *d_tx_ = nullptr, *d_bits_ = nullptr;
  float *d_llrs_ = nullptr;
  int32_t *d_its_ = nullptr;
  unsigned long long *d_counters_ = nullptr;
  size_t cap_frames_ = 0, cap_llr_rows_ = 0;
  uint64_t streamed_frames_ = 0;
  // continuous batching is built and exact but does not pay in this data layout (profiles/r03_continuous_batching.txt:
  // 0.64-0.69 of the iteration-proportional bound against 0.75-0.81 for drained batches with compaction): opt-in
  bool streaming_ = false;
  // Straggler pooling: once a run() call has seen how many iterations its frames take, later chunks run a reduced
  // budget and the frames that have not converged by then are pooled and decoded together with the full budget --
  // instead of every chunk dragging its few slow (or failing) frames through launch-bound, nearly empty iterations
  // up to max_iterations.  Per frame the result is the one of a single full-budget decode (the decoder is
  // deterministic per frame and does not depend on the batch around it): identical counters.
  bool pooling_ = true;
  uint64_t pooled_frames_ = 0;
  bool budget_valid_ = false;  // the reduced budget the last call arrived at, and the point it belongs to
  uint32_t budget_ = 0, budget_max_it_ = 0;
  double budget_ebn0_ = 0.0;
  float *d_pool_llrs_ = nullptr;
  uint64_t *d_pool_frames_ = nullptr;
  uint32_t *d_pool_count_ = nullptr;
  uint8_t *d_pool_bits_ = nullptr;
  int32_t *d_pool_its_ = nullptr;
  size_t pool_cap_ = 0;
  int ensure_pool(size_t capacity);
  int flush_pool(uint32_t count, uint64_t seed, uint32_t max_iterations, uint64_t bch_max_errors, size_t chunk_group);
  hipStream_t stream_ = nullptr;
  std::string error_;
};

}  // namespace ldpc
