#include "simulator.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "frame_gen.hip.h"
#include "implementation.h"
#include "sparse.h"

namespace ldpc {

bool Simulator::fail(const std::string &m, hipError_t e) {
  error_ = m;
  if (e != hipSuccess) error_ += std::string(": ") + hipGetErrorString(e);
  std::fprintf(stderr, "ldpc_toolbox (hip): simulator: %s\n", error_.c_str());
  return false;
}

#define SIM_TRY(expr)                 \
  do {                                \
    hipError_t _e = (expr);           \
    if (_e != hipSuccess) {           \
      fail(#expr, _e);                \
      return -2;                      \
    }                                 \
  } while (0)

Simulator *Simulator::create(const std::string &alist, const std::string &implementation,
                             const std::string &puncturing, int device, uint32_t pool, uint64_t pool_seed,
                             std::string *err) {
  auto bail = [&](const std::string &m) -> Simulator * {
    if (err) *err = m;
    return nullptr;
  };
  SparseMatrix h;
  std::string e;
  if (!SparseMatrix::from_alist(alist, &h, &e)) return bail(e);
  Implementation impl;
  if (!parse_implementation(implementation, &impl, &e)) return bail(e);
  std::vector<uint8_t> pattern;
  if (!parse_puncturing_pattern(puncturing, &pattern)) return bail("invalid puncturing pattern");
  Encoder enc;
  if (!Encoder::from_h(h, &enc, &e)) return bail(e);
  std::unique_ptr<Simulator> s(new Simulator());
  s->dec_.reset(DeviceDecoder::create(h, impl, pattern, device, &e));
  if (!s->dec_) return bail(e);
  // (run() reads the counters back right after every chunk: the decode calls may as well pace themselves on the
  // groups' progress instead of enqueuing every iteration up to max_iterations)
  (void)s->dec_->set_option("throttle", 1);
  s->device_ = device;
  s->n_ = h.num_cols();
  s->k_ = h.num_cols() - h.num_rows();
  s->n_tx_ = s->dec_->input_len();
  s->pool_ = std::max<uint32_t>(pool, 1);
  // the pool: random messages, systematic encode, puncture (puncturing.rs:47-75)
  std::mt19937_64 rng(pool_seed);
  s->messages_.resize(size_t(s->pool_) * s->k_);
  s->tx_bits_.resize(size_t(s->pool_) * s->n_tx_);
  std::vector<uint8_t> cw(s->n_);
  for (uint32_t p = 0; p < s->pool_; p++) {
    uint8_t *m = &s->messages_[size_t(p) * s->k_];
    for (size_t i = 0; i < s->k_; i += 64) {
      uint64_t w = rng();
      for (size_t j = i; j < std::min(i + 64, s->k_); j++, w >>= 1) m[j] = w & 1;
    }
    enc.encode(m, cw.data());
    uint8_t *tx = &s->tx_bits_[size_t(p) * s->n_tx_];
    if (pattern.empty()) {
      std::copy(cw.begin(), cw.end(), tx);
    } else {
      const size_t block = s->n_ / pattern.size();
      size_t j = 0;
      for (size_t b = 0; b < pattern.size(); b++)
        if (pattern[b]) std::copy(cw.begin() + b * block, cw.begin() + (b + 1) * block, tx + (j++) * block);
    }
  }
  if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice failed");
  bool ok = hipMalloc(reinterpret_cast<void **>(&s->d_messages_), s->messages_.size()) == hipSuccess &&
            hipMalloc(reinterpret_cast<void **>(&s->d_tx_), s->tx_bits_.size()) == hipSuccess &&
            hipMalloc(reinterpret_cast<void **>(&s->d_counters_), 9 * sizeof(unsigned long long)) == hipSuccess &&
            hipMemcpy(s->d_messages_, s->messages_.data(), s->messages_.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(s->d_tx_, s->tx_bits_.data(), s->tx_bits_.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipStreamCreateWithFlags(&s->stream_, hipStreamNonBlocking) == hipSuccess;
  if (!ok) return bail("device allocation for the simulator failed");
  return s.release();
}

Simulator::~Simulator() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  for (void *p : {(void *)d_messages_, (void *)d_tx_, (void *)d_bits_, (void *)d_llrs_, (void *)d_its_,
                  (void *)d_counters_, (void *)d_pool_llrs_, (void *)d_pool_frames_, (void *)d_pool_count_,
                  (void *)d_pool_bits_, (void *)d_pool_its_})
    if (p) (void)hipFree(p);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// device buffers: LLR rows (one group's worth is enough for the streamed chunks), decoded bits and iteration counts
int Simulator::ensure(size_t frames, size_t llr_rows) {
  if (llr_rows > cap_llr_rows_) {
    if (d_llrs_) (void)hipFree(d_llrs_);
    d_llrs_ = nullptr;
    cap_llr_rows_ = 0;
    SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_llrs_), llr_rows * n_tx_ * sizeof(float)));
    cap_llr_rows_ = llr_rows;
  }
  if (frames <= cap_frames_) return 0;
  for (void *p : {(void *)d_bits_, (void *)d_its_})
    if (p) (void)hipFree(p);
  d_bits_ = nullptr;
  d_its_ = nullptr;
  cap_frames_ = 0;
  SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_bits_), frames * std::max<size_t>(k_, 1)));
  SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_its_), frames * sizeof(int32_t)));
  cap_frames_ = frames;
  return 0;
}

int Simulator::ensure_pool(size_t capacity) {
  if (capacity <= pool_cap_) return 0;
  for (void *p : {(void *)d_pool_llrs_, (void *)d_pool_frames_, (void *)d_pool_count_, (void *)d_pool_bits_, (void *)d_pool_its_})
    if (p) (void)hipFree(p);
  d_pool_llrs_ = nullptr;
  d_pool_frames_ = nullptr;
  d_pool_count_ = nullptr;
  d_pool_bits_ = nullptr;
  d_pool_its_ = nullptr;
  pool_cap_ = 0;
  SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_pool_llrs_), capacity * n_tx_ * sizeof(float)));
  SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_pool_frames_), capacity * sizeof(uint64_t)));
  SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_pool_count_), 64));
  SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_pool_bits_), capacity * std::max<size_t>(k_, 1)));
  SIM_TRY(hipMalloc(reinterpret_cast<void **>(&d_pool_its_), capacity * sizeof(int32_t)));
  pool_cap_ = capacity;
  return 0;
}

// the pooled stragglers, with the full iteration budget (the stream is idle: the caller has just read the count)
int Simulator::flush_pool(uint32_t count, uint64_t seed, uint32_t max_iterations, uint64_t bch_max_errors,
                          size_t chunk_group) {
  // (a call on the decoder's own stream: a handful of stragglers takes the single-launch small-batch path; a pool too
  // large for that is decoded in the chunks' group size, so that the decoder keeps the workspace it has -- a group
  // sized by the pool would free and re-allocate gigabytes before and after every flush)
  dec_->set_min_group(chunk_group);
  const int drc = dec_->decode_device(d_pool_llrs_, false, count, max_iterations, d_pool_bits_, k_, d_pool_its_, nullptr, nullptr);
  dec_->set_min_group(0);
  if (drc) {
    error_ = dec_->last_error();
    return drc;
  }
  gen::count_errors_kernel<<<(count * 64 + 255) / 256, 256, 0, stream_>>>(
      d_pool_bits_, static_cast<uint32_t>(k_), d_pool_its_, d_messages_, static_cast<uint32_t>(k_), pool_, seed, 0, count,
      max_iterations, bch_max_errors, d_counters_, d_pool_frames_, nullptr, 0);
  SIM_TRY(hipMemsetAsync(d_pool_count_, 0, sizeof(uint32_t), stream_));
  pooled_frames_ += count;
  return 0;
}

bool Simulator::set_modulation(int bits_per_symbol) {
  if (bits_per_symbol == 1 || (bits_per_symbol == 3 && n_tx_ % 3 == 0)) {
    bits_per_symbol_ = bits_per_symbol;
    budget_valid_ = false;  // (the iteration budget of straggler pooling belongs to one channel set-up)
    return true;
  }
  fail(bits_per_symbol == 3 ? "8PSK needs a transmitted length that is a multiple of 3 (modulation.rs:188-193)"
                            : "modulation must be 1 (BPSK) or 3 (8PSK)");
  return false;
}

bool Simulator::set_interleaving(int64_t columns) {
  const uint64_t c = columns < 0 ? uint64_t(-columns) : uint64_t(columns);
  if (c > 0x7FFFFFFFull || (c != 0 && n_tx_ % c != 0)) {
    fail("interleaver columns must divide the transmitted length (interleaving.rs:44)");
    return false;
  }
  interleaving_ = columns;
  budget_valid_ = false;
  return true;
}

// ber.rs:299-302: EsN0 = rate * bits_per_symbol * EbN0, sigma = sqrt(0.5 / EsN0)
double Simulator::noise_sigma(double ebn0_db) const {
  const double ebn0 = std::pow(10.0, 0.1 * ebn0_db);
  return std::sqrt(0.5 / (rate() * static_cast<double>(bits_per_symbol_) * ebn0));
}

// BPSK: the f32 values the generator uses are the roundings of these doubles
void Simulator::noise_params(double ebn0_db, float *sigma, float *scale) const {
  const double s = noise_sigma(ebn0_db);
  *sigma = static_cast<float>(s);
  *scale = static_cast<float>(-2.0 / (s * s));
}

// frames [first_frame, first_frame + frames) -> d_llrs_ (codeword order, ready for the decoder)
void Simulator::launch_generator(double ebn0_db, uint64_t seed, uint64_t first_frame, uint32_t frames, float *dst) {
  const uint32_t n_tx = static_cast<uint32_t>(n_tx_);
  if (dst == nullptr) dst = d_llrs_;
  if (bits_per_symbol_ == 3) {
    const double s = noise_sigma(ebn0_db);
    const uint64_t threads = uint64_t(frames) * (n_tx / 3);
    gen::psk8_llr_kernel<<<static_cast<uint32_t>((threads + 255) / 256), 256, 0, stream_>>>(
        d_tx_, pool_, n_tx, static_cast<int32_t>(interleaving_), seed, first_frame, frames, s, 1.0 / (s * s), dst);
    return;
  }
  // BPSK: one LLR per bit, so interleaving followed by deinterleaving changes nothing but which
  // noise sample a position gets; the generator keys the noise by codeword position
  float sigma, scale;
  noise_params(ebn0_db, &sigma, &scale);
  const uint64_t threads = uint64_t(frames) * ((n_tx + 1) / 2);
  gen::awgn_llr_kernel<<<static_cast<uint32_t>((threads + 255) / 256), 256, 0, stream_>>>(
      d_tx_, pool_, n_tx, seed, first_frame, frames, sigma, scale, dst);
}

int Simulator::run(double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames, uint32_t max_iterations,
                   uint64_t counters[6]) {
  uint64_t all[9];
  const int rc = run_bch(ebn0_db, seed, first_frame, frames, max_iterations, 0, all);
  for (int i = 0; i < 6; i++) counters[i] = all[i];
  return rc;
}

int Simulator::run_bch(double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames, uint32_t max_iterations,
                       uint64_t bch_max_errors, uint64_t counters[9]) {
  for (int i = 0; i < 9; i++) counters[i] = 0;
  if (frames == 0) return 0;
  SIM_TRY(hipSetDevice(device_));
  // Continuous batching where the decoder offers it (flooding Minsumf32, BPSK): the frames of a chunk are produced
  // on demand, straight into the slots that finished codewords free (DeviceDecoder::decode_stream); the chip stays
  // full over the whole chunk instead of draining every 4096 frames.  Frame f is the same frame either way.
  const bool streaming = streaming_ && stream_ && dec_->stream_capable() && bits_per_symbol_ == 1 && max_iterations > 0 &&
                         frames > dec_->stream_group() && std::getenv("LDPC_TOOLBOX_NO_STREAM") == nullptr;
  streamed_frames_ = 0;
  // (drained path: a chunk is one group of the decoder -- 4096 frames, more for small graphs)
  const size_t chunk = streaming ? std::min<size_t>(frames, 32768) : std::min<size_t>(frames, std::max<size_t>(dec_->preferred_group(frames), 4096));
  if (int rc = ensure(chunk, streaming ? std::min<size_t>(chunk, 4096) : chunk)) return rc;
  SIM_TRY(hipMemsetAsync(d_counters_, 0, 9 * sizeof(unsigned long long), stream_));
  // straggler pooling (simulator.h): from the second chunk on, when the frames seen so far converge well within the budget
  const bool track = pooling_ && !streaming && max_iterations >= 24;
  const bool can_pool = track && (frames > chunk || (budget_valid_ && budget_ebn0_ == ebn0_db && budget_max_it_ == max_iterations));
  pooled_frames_ = 0;
  if (can_pool) {
    if (int rc = ensure_pool(2 * chunk)) return rc;
    SIM_TRY(hipMemsetAsync(d_pool_count_, 0, sizeof(uint32_t), stream_));
  }
  // (a sweep calls run() again and again at one Eb/N0: the budget the previous call arrived at carries over)
  const bool same_point = can_pool && budget_valid_ && budget_ebn0_ == ebn0_db && budget_max_it_ == max_iterations;
  uint32_t budget = same_point ? std::min(budget_, max_iterations) : max_iterations;
  uint32_t next_call_budget = budget;
  const size_t chunk_group = dec_->preferred_group(chunk);
  // The next chunk's budget from the counters so far: twice the average iteration count of the frames that converge,
  // plus 8 -- where a frame still waiting in the pool counts with the budget it has exhausted (a lower bound; left out,
  // the average would cover only the frames that beat the budget and ratchet it down) -- and the full budget again when
  // more than a quarter of the call's frames have needed the second pass (pooled now, flushed earlier, or failed
  // outright in a full-budget chunk): then pooling decodes too much twice.
  auto next_budget = [&](const unsigned long long c[9], uint32_t pooled, uint32_t used_budget, uint32_t max_it) -> uint32_t {
    const double counted = static_cast<double>(c[0]), clean = counted - static_cast<double>(c[2]);
    const double total_its_failed = static_cast<double>(c[4]) - static_cast<double>(c[5]);
    const double failed_full = std::min(static_cast<double>(c[2]), total_its_failed / std::max<double>(max_it, 1));
    const double second_pass = static_cast<double>(pooled) + static_cast<double>(pooled_frames_) + failed_full;
    const double ok_frames = clean + pooled, ok_its = static_cast<double>(c[5]) + static_cast<double>(pooled) * used_budget;
    const double avg_ok = ok_frames > 0 ? ok_its / ok_frames : static_cast<double>(max_it);
    uint32_t next = static_cast<uint32_t>(std::min<double>(max_it, std::ceil(2.0 * avg_ok) + 8.0));
    next = std::max<uint32_t>(next, 16);
    if (second_pass > 0.25 * (counted + pooled) || next * 10 >= max_it * 7) next = max_it;  // nothing to gain
    return next;
  };
  for (size_t f0 = 0; f0 < frames; f0 += chunk) {
    const uint32_t nf = static_cast<uint32_t>(std::min(chunk, frames - f0));
    if (streaming) {
      float sigma, scale;
      noise_params(ebn0_db, &sigma, &scale);
      const uint64_t base = first_frame + f0;
      const uint32_t n_tx = static_cast<uint32_t>(n_tx_);
      auto source = [&](const uint64_t *first_count, float *dst, hipStream_t s) {
        gen::awgn_llr_stream_kernel<<<8192, 256, 0, s>>>(d_tx_, pool_, n_tx, seed, base, first_count, sigma, scale, dst);
      };
      SIM_TRY(hipStreamSynchronize(stream_));  // (decode_stream runs on the decoder's own stream)
      if (int rc = dec_->decode_stream(source, d_llrs_, nf, max_iterations, d_bits_, k_, d_its_)) {
        error_ = dec_->last_error();
        return rc;
      }
      streamed_frames_ += nf;
    } else {
    launch_generator(ebn0_db, seed, first_frame + f0, nf);
    if (int rc = dec_->decode_device(d_llrs_, false, nf, budget, d_bits_, k_, d_its_, nullptr, stream_)) {
      error_ = dec_->last_error();
      return rc;
    }
    }
    const bool reduced = budget < max_iterations;
    if (reduced)
      gen::straggler_collect_kernel<<<(nf * 64 + 255) / 256, 256, 0, stream_>>>(
          d_its_, nf, first_frame + f0, d_llrs_, static_cast<uint32_t>(n_tx_), d_pool_llrs_, d_pool_frames_, d_pool_count_,
          static_cast<uint32_t>(pool_cap_));
    gen::count_errors_kernel<<<(nf * 64 + 255) / 256, 256, 0, stream_>>>(
        d_bits_, static_cast<uint32_t>(k_), d_its_, d_messages_, static_cast<uint32_t>(k_), pool_, seed,
        first_frame + f0, nf, max_iterations, bch_max_errors, d_counters_, nullptr, nullptr, reduced ? 1 : 0);
    if (can_pool) {
      // what the frames of this call have needed so far decides the next chunk's budget (a call that cannot pool -- a
      // single chunk at a new point -- reads the counters once, at the end: no synchronisation between its launches)
      unsigned long long c[9];
      uint32_t pooled = 0;
      SIM_TRY(hipMemcpyAsync(c, d_counters_, sizeof(c), hipMemcpyDeviceToHost, stream_));
      SIM_TRY(hipMemcpyAsync(&pooled, d_pool_count_, sizeof(pooled), hipMemcpyDeviceToHost, stream_));
      SIM_TRY(hipStreamSynchronize(stream_));
      if (pooled > pool_cap_) {
        fail("straggler pool overflow");
        return -3;
      }
      const uint32_t next = next_budget(c, pooled, budget, max_iterations);
      budget = next;
      next_call_budget = next;
      if (f0 + chunk < frames && pooled + chunk > pool_cap_) {
        if (int rc = flush_pool(pooled, seed, max_iterations, bch_max_errors, chunk_group)) return rc;
      }
    }
  }
  if (can_pool) {
    budget_valid_ = true;
    budget_ = next_call_budget;
    budget_ebn0_ = ebn0_db;
    budget_max_it_ = max_iterations;
  }
  if (can_pool) {
    uint32_t pooled = 0;
    SIM_TRY(hipMemcpyAsync(&pooled, d_pool_count_, sizeof(pooled), hipMemcpyDeviceToHost, stream_));
    SIM_TRY(hipStreamSynchronize(stream_));
    if (pooled > pool_cap_) {
      fail("straggler pool overflow");
      return -3;
    }
    if (pooled)
      if (int rc = flush_pool(pooled, seed, max_iterations, bch_max_errors, chunk_group)) return rc;
  }
  unsigned long long host[9];
  SIM_TRY(hipMemcpyAsync(host, d_counters_, sizeof(host), hipMemcpyDeviceToHost, stream_));
  SIM_TRY(hipStreamSynchronize(stream_));
  SIM_TRY(hipGetLastError());
  if (track && !can_pool) {  // every frame of this call ran the full budget: what they needed is the next call's estimate
    budget_valid_ = true;
    budget_ = next_budget(host, 0, max_iterations, max_iterations);
    budget_ebn0_ = ebn0_db;
    budget_max_it_ = max_iterations;
  }
  for (int i = 0; i < 9; i++) counters[i] = host[i];
  return 0;
}

int Simulator::generate(double ebn0_db, uint64_t seed, uint64_t first_frame, size_t frames, float *llrs,
                        uint32_t *pool_index) {
  if (frames == 0) return 0;
  SIM_TRY(hipSetDevice(device_));
  // `llrs` in this GPU's memory: the frames are generated in place (a multi-rank benchmark fills its gigabyte of frames
  // without a round trip through pageable host memory); anywhere else: generated here and copied out
  hipPointerAttribute_t attr{};
  const bool on_device = hipPointerGetAttributes(&attr, llrs) == hipSuccess && attr.type == hipMemoryTypeDevice && attr.device == device_;
  (void)hipGetLastError();  // (an ordinary host pointer is "invalid value" to the query on some runtimes)
  if (on_device) {
    launch_generator(ebn0_db, seed, first_frame, static_cast<uint32_t>(frames), llrs);
  } else {
    if (int rc = ensure(0, frames)) return rc;
    launch_generator(ebn0_db, seed, first_frame, static_cast<uint32_t>(frames));
    SIM_TRY(hipMemcpyAsync(llrs, d_llrs_, frames * n_tx_ * sizeof(float), hipMemcpyDefault, stream_));
  }
  SIM_TRY(hipStreamSynchronize(stream_));
  SIM_TRY(hipGetLastError());
  if (pool_index)
    for (size_t f = 0; f < frames; f++) pool_index[f] = gen::pool_index(seed, first_frame + f, pool_);
  return 0;
}

}  // namespace ldpc
