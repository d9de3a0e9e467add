// On-device frame generation and error counting around the decode path -- the GPU counterpart of
// the reference's per-frame pipeline Worker::simulate (/root/reference/src/simulation/ber.rs:436-481):
// codeword -> puncture -> BPSK (bit 1 -> +1, bit 0 -> -1, modulation.rs:87-95) -> AWGN
// (channel.rs:60-81) -> LLR = -2 y / sigma^2 (modulation.rs:127-140) -> [decode] -> bit errors on
// the first k bits, frame error, false decode, iterations (ber.rs:468-480).
//
// The reference draws from an OS-seeded ThreadRng, so only the *distribution* can be matched.
// Here the noise is a pure function of (seed, frame index, position): Philox4x32-10 counters and the
// Marsaglia polar method in f32, using only operations that are correctly rounded on both sides
// (+, *, /, sqrt) and the glibc-identical logf of exact_math.h -- so a CPU restatement regenerates
// the very same frames bit for bit (the test suite does exactly that).
//
// 8PSK (modulation.rs:144-288) with the DVB-S2 bit interleaver (interleaving.rs:20-87) is the
// other modulation of the reference's driver: interleave -> Gray-mapped 8PSK -> complex AWGN ->
// exact max* demodulation -> deinterleave.  From the noisy symbol on, the arithmetic is the
// reference's f64 arithmetic (exp / ln_1p through the glibc-identical exact_math.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "exact_math.h"

namespace ldpc {
namespace gen {

#if defined(__HIPCC__) || defined(__HIP__)
#define GEN_FN __host__ __device__ __forceinline__
#else
#define GEN_FN static inline
#endif

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
GEN_FN void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = uint64_t(0xD2511F53u) * c[0];
    const uint64_t p1 = uint64_t(0xCD9E8D57u) * c[2];
    const uint32_t n0 = uint32_t(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = uint32_t(p1);
    const uint32_t n2 = uint32_t(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = uint32_t(p0);
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// 24-bit uniform in [-1, 1): exact in f32
GEN_FN float unit(uint32_t w) { return static_cast<float>(w >> 8) * 0x1p-23f - 1.0f; }

// Two standard normals for the position pair `pair` of frame `frame`: polar method over
// successive Philox blocks (counter word 0 = attempt); each block offers two candidate points.
GEN_FN void normal_pair(uint64_t seed, uint64_t frame, uint32_t pair, float *z0, float *z1) {
  for (uint32_t attempt = 0;; attempt++) {
    uint32_t c[4] = {attempt, pair, uint32_t(frame), uint32_t(frame >> 32)};
    philox4x32_10(c, uint32_t(seed), uint32_t(seed >> 32));
    for (int h = 0; h < 2; h++) {
      const float v1 = unit(c[2 * h]), v2 = unit(c[2 * h + 1]);
      const float s = v1 * v1 + v2 * v2;
      if (s > 0.0f && s < 1.0f) {
        const float f = __builtin_sqrtf(-2.0f * em::logf(s) / s);
        *z0 = v1 * f;
        *z1 = v2 * f;
        return;
      }
    }
  }
}

// which pooled codeword frame `frame` transmits
GEN_FN uint32_t pool_index(uint64_t seed, uint64_t frame, uint32_t pool) {
  uint32_t c[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, uint32_t(frame), uint32_t(frame >> 32)};
  philox4x32_10(c, uint32_t(seed), uint32_t(seed >> 32));
  return c[0] % pool;
}

// LLR of transmitted position j of a frame whose (punctured) codeword bit is `bit`
GEN_FN float llr_from(uint32_t bit, float z, float sigma, float scale) {
  const float sym = bit ? 1.0f : -1.0f;
  const float y = sym + sigma * z;
  return scale * y;  // scale = -2 / sigma^2
}

// ---- 8PSK -----------------------------------------------------------------------------------
// interleaving.rs:40-58: the codeword is written row-wise into a [columns][rows] array and read
// column-wise (the rows optionally backwards): interleaved position i = r * columns + c' holds
// codeword position c * rows + r with c = backwards ? columns - 1 - c' : c'.  Deinterleaving the
// LLRs (interleaving.rs:65-86) is the inverse, so the LLR of interleaved position i belongs to
// that same codeword position.  columns == 0: no interleaver.
GEN_FN uint32_t deinterleaved_position(uint32_t i, uint32_t n_tx, int32_t interleaving) {
  if (interleaving == 0) return i;
  const uint32_t columns = interleaving < 0 ? uint32_t(-interleaving) : uint32_t(interleaving);
  const uint32_t rows = n_tx / columns;
  const uint32_t r = i / columns, cp = i % columns;
  const uint32_t c = interleaving < 0 ? columns - 1 - cp : cp;
  return c * rows + r;
}

// modulation.rs:166-179: the DVB-S2 Gray-coded constellation, (b0, b1, b2) -> point
GEN_FN void psk8_point(uint32_t b0, uint32_t b1, uint32_t b2, double *re, double *im) {
  const double a = 0.70710678118654757;  // (0.5f64).sqrt()
  const uint32_t key = b0 | (b1 << 1) | (b2 << 2);
  switch (key) {
    case 0: *re = a; *im = a; break;        // 000
    case 1: *re = 0.0; *im = 1.0; break;    // 100
    case 3: *re = -a; *im = a; break;       // 110
    case 2: *re = -1.0; *im = 0.0; break;   // 010
    case 6: *re = -a; *im = -a; break;      // 011
    case 7: *re = 0.0; *im = -1.0; break;   // 111
    case 5: *re = a; *im = -a; break;       // 101
    default: *re = 1.0; *im = 0.0; break;   // 001
  }
}

// modulation.rs:286-288
GEN_FN double maxstar(double a, double b) {
  const double d = a - b;
  return (a > b ? a : b) + em::log1p(em::exp(-(d < 0.0 ? -d : d)));
}

// modulation.rs:225-267: exact bit LLRs of one received symbol; scale = 1 / sigma^2
GEN_FN void psk8_demodulate(double re, double im, double scale, double llr[3]) {
  const double a = 0.70710678118654757;
  re = re * scale;
  im = im * scale;
  const double d000 = re * a + im * a;
  const double d100 = re * 0.0 + im * 1.0;
  const double d110 = re * -a + im * a;
  const double d010 = re * -1.0 + im * 0.0;
  const double d011 = re * -a + im * -a;
  const double d111 = re * 0.0 + im * -1.0;
  const double d101 = re * a + im * -a;
  const double d001 = re * 1.0 + im * 0.0;
  llr[0] = maxstar(maxstar(maxstar(d000, d001), d010), d011) - maxstar(maxstar(maxstar(d100, d101), d110), d111);
  llr[1] = maxstar(maxstar(maxstar(d000, d001), d100), d101) - maxstar(maxstar(maxstar(d010, d011), d110), d111);
  llr[2] = maxstar(maxstar(maxstar(d000, d010), d100), d110) - maxstar(maxstar(maxstar(d001, d011), d101), d111);
}

// The three LLRs of symbol `sym` of a frame: the symbol carries interleaved positions 3 sym .. 3 sym + 2,
// the noise is (sigma z0, sigma z1) with the normal pair of index `sym`.  pos[b] = codeword position
// (of the punctured codeword) the b-th LLR belongs to.
GEN_FN void psk8_symbol_llrs(const uint8_t *cw, uint32_t n_tx, int32_t interleaving, uint64_t seed, uint64_t frame,
                             uint32_t sym, double sigma, double scale, uint32_t pos[3], float out[3]) {
  float z0, z1;
  normal_pair(seed, frame, sym, &z0, &z1);
  uint32_t bit[3];
  for (int b = 0; b < 3; b++) {
    pos[b] = deinterleaved_position(3 * sym + b, n_tx, interleaving);
    bit[b] = cw[pos[b]] & 1u;
  }
  double re, im, llr[3];
  psk8_point(bit[0], bit[1], bit[2], &re, &im);
  re = re + sigma * static_cast<double>(z0);   // channel.rs:76-81
  im = im + sigma * static_cast<double>(z1);
  psk8_demodulate(re, im, scale, llr);
  for (int b = 0; b < 3; b++) out[b] = static_cast<float>(llr[b]);
}

#if defined(__HIPCC__) || defined(__HIP__)
// 8PSK frames: one thread per symbol; llrs [frames][n_tx] f32 in codeword (deinterleaved) order
__global__ __launch_bounds__(256) void psk8_llr_kernel(const uint8_t *__restrict__ tx_bits, uint32_t pool,
                                                       uint32_t n_tx, int32_t interleaving, uint64_t seed,
                                                       uint64_t first_frame, uint32_t frames, double sigma,
                                                       double scale, float *__restrict__ llrs) {
  const uint32_t symbols = n_tx / 3;
  const uint64_t id = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= uint64_t(frames) * symbols) return;
  const uint32_t f = static_cast<uint32_t>(id / symbols), sym = static_cast<uint32_t>(id % symbols);
  const uint64_t frame = first_frame + f;
  const uint8_t *cw = tx_bits + size_t(pool_index(seed, frame, pool)) * n_tx;
  uint32_t pos[3];
  float out[3];
  psk8_symbol_llrs(cw, n_tx, interleaving, seed, frame, sym, sigma, scale, pos, out);
  float *row = llrs + size_t(f) * n_tx;
  for (int b = 0; b < 3; b++) row[pos[b]] = out[b];
}

// llrs [frames][n_tx] f32; tx_bits [pool][n_tx] u8 (punctured codewords); one thread per pair
__global__ __launch_bounds__(256) void awgn_llr_kernel(const uint8_t *__restrict__ tx_bits, uint32_t pool,
                                                       uint32_t n_tx, uint64_t seed, uint64_t first_frame,
                                                       uint32_t frames, float sigma, float scale,
                                                       float *__restrict__ llrs) {
  const uint32_t pairs = (n_tx + 1) / 2;
  const uint64_t id = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= uint64_t(frames) * pairs) return;
  const uint32_t f = static_cast<uint32_t>(id / pairs), pair = static_cast<uint32_t>(id % pairs);
  const uint64_t frame = first_frame + f;
  float z0, z1;
  normal_pair(seed, frame, pair, &z0, &z1);
  const uint8_t *cw = tx_bits + size_t(pool_index(seed, frame, pool)) * n_tx;
  float *row = llrs + size_t(f) * n_tx;
  const uint32_t j = 2 * pair;
  row[j] = llr_from(cw[j], z0, sigma, scale);
  if (j + 1 < n_tx) row[j + 1] = llr_from(cw[j + 1], z1, sigma, scale);
}

// The same frames on demand (continuous batching, DeviceDecoder::decode_stream): frames base_frame + first ..
// + first + count - 1, with first = first_count[0] and count = first_count[1] read from device memory (a harvest of
// the decoder decides them); rows [count][n_tx].  Grid-stride: the launch does not know the count.
__global__ __launch_bounds__(256) void awgn_llr_stream_kernel(const uint8_t *__restrict__ tx_bits, uint32_t pool,
                                                              uint32_t n_tx, uint64_t seed, uint64_t base_frame,
                                                              const uint64_t *__restrict__ first_count, float sigma,
                                                              float scale, float *__restrict__ llrs) {
  const uint32_t pairs = (n_tx + 1) / 2;
  const uint64_t first = first_count[0], total = first_count[1] * pairs;
  for (uint64_t id = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; id < total; id += uint64_t(gridDim.x) * blockDim.x) {
    const uint32_t f = static_cast<uint32_t>(id / pairs), pair = static_cast<uint32_t>(id % pairs);
    const uint64_t frame = base_frame + first + f;
    float z0, z1;
    normal_pair(seed, frame, pair, &z0, &z1);
    const uint8_t *cw = tx_bits + size_t(pool_index(seed, frame, pool)) * n_tx;
    float *row = llrs + size_t(f) * n_tx;
    const uint32_t j = 2 * pair;
    row[j] = llr_from(cw[j], z0, sigma, scale);
    if (j + 1 < n_tx) row[j + 1] = llr_from(cw[j + 1], z1, sigma, scale);
  }
}

// counters (ber.rs:113-138, 318-338): [0] frames [1] bit_errors [2] frame_errors [3] false_decodes
// [4] total_iterations [5] correct_iterations, and the outer-BCH view of the same frames
// (ber.rs:328-337: a frame with at most bch_max_errors bit errors counts as corrected):
// [6] bch bit_errors [7] bch frame_errors [8] bch correct_iterations (left at zero when
// bch_max_errors == 0).  One wave per frame: lanes stride over the k message bits, a shuffle
// reduction adds the lane counts, lane 0 does the 64-bit atomics.
__global__ __launch_bounds__(256) void count_errors_kernel(const uint8_t *__restrict__ decoded, uint32_t out_len,
                                                           const int32_t *__restrict__ iterations,
                                                           const uint8_t *__restrict__ messages, uint32_t k,
                                                           uint32_t pool, uint64_t seed, uint64_t first_frame,
                                                           uint32_t frames, uint32_t max_iterations,
                                                           uint64_t bch_max_errors,
                                                           unsigned long long *__restrict__ counters,
                                                           const uint64_t *__restrict__ frame_ids = nullptr,
                                                           const uint32_t *__restrict__ frame_count = nullptr,
                                                           int skip_failed = 0) {
  // frame_ids / frame_count: the frames are the pooled stragglers (their numbers and how many, in device memory);
  // skip_failed: frames that did not converge are not counted here (they went to the pool: straggler_collect_kernel)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t f = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (f >= (frame_count ? min(*frame_count, frames) : frames)) return;
  if (skip_failed && iterations[f] < 0) return;
  const uint8_t *msg = messages + size_t(pool_index(seed, frame_ids ? frame_ids[f] : first_frame + f, pool)) * k;
  const uint8_t *dec = decoded + size_t(f) * out_len;
  uint32_t errs = 0;
  for (uint32_t i = lane; i < k; i += 64) errs += (dec[i] != msg[i]) ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) errs += __shfl_down(errs, off, 64);
  if (lane == 0) {
    const int32_t it = iterations[f];
    const bool success = it >= 0;
    const uint32_t its = success ? static_cast<uint32_t>(it) : max_iterations;
    atomicAdd(&counters[0], 1ull);
    if (errs) {
      atomicAdd(&counters[1], static_cast<unsigned long long>(errs));
      atomicAdd(&counters[2], 1ull);
      if (success) atomicAdd(&counters[3], 1ull);
    } else {
      atomicAdd(&counters[5], static_cast<unsigned long long>(its));
    }
    atomicAdd(&counters[4], static_cast<unsigned long long>(its));
    if (bch_max_errors > 0) {
      if (errs > bch_max_errors) {
        atomicAdd(&counters[6], static_cast<unsigned long long>(errs));
        atomicAdd(&counters[7], 1ull);
      } else {
        atomicAdd(&counters[8], static_cast<unsigned long long>(its));
      }
    }
  }
}

// Straggler pooling (Simulator::run_bch): the frames of a chunk that have not converged within the chunk's iteration
// budget are set aside -- their LLR rows and frame numbers appended to a pool -- and decoded later, many at a time, with
// the full budget.  One wavefront per frame.
__global__ __launch_bounds__(256) void straggler_collect_kernel(const int32_t *__restrict__ iterations, uint32_t frames,
                                                                uint64_t first_frame, const float *__restrict__ llrs,
                                                                uint32_t n_tx, float *__restrict__ pool_llrs,
                                                                uint64_t *__restrict__ pool_frames,
                                                                uint32_t *__restrict__ pool_count, uint32_t capacity) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t f = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (f >= frames || iterations[f] >= 0) return;
  uint32_t pos = 0;
  if (lane == 0) pos = atomicAdd(pool_count, 1u);
  pos = static_cast<uint32_t>(__shfl(static_cast<int>(pos), 0, 64));
  if (pos >= capacity) return;  // (the host keeps a chunk's worth of room: cannot happen; the count then shows it)
  if (lane == 0) pool_frames[pos] = first_frame + f;
  const float *src = llrs + size_t(f) * n_tx;
  float *dst = pool_llrs + size_t(pos) * n_tx;
  for (uint32_t i = lane; i < n_tx; i += 64) dst[i] = src[i];
}
#endif

}  // namespace gen
}  // namespace ldpc
