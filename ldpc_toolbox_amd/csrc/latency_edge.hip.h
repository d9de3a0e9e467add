// Small-batch (latency) path with the lanes across the EDGES of one codeword: the horizontal-layered schedule
// (every float rule) and the flooding schedule of the sum-product family (Phi, Tanh, Minstarapprox, Aminstar; flooding
// min-sum in f32 has the row-lane kernel of latency.hip.h), f32 and f64 arithmetic, and the 8-bit rules on both schedules.  The reference's own call pattern
// is ONE codeword per decode call (/root/reference/src/c_api/decoder.rs:50-67, src/simulation/ber.rs:462-466) -- and
// the decoder its command line defaults to is flooding Phif64 (src/cli/ber.rs:49) -- which through the batched kernels
// costs one launch per dependency level or two per flooding iteration with 1 lane in 64 useful.  Here, as in
// latency.hip.h:
//
//   * one persistent launch per call; a codeword is owned by one XCD (its soft values and messages stay in that
//     XCD's L2); up to 8 codewords of a call decode concurrently, one per XCD, and larger calls give every XCD a
//     BUNDLE of up to 8 codewords that share each phase and each barrier (64 codewords in about the time of 8);
//   * a LANE owns one EDGE of one row.  Rows are packed, whole, into wavefront-sized chunks of at most 64 edge
//     lanes; a row's d lanes sit next to each other in one wavefront and exchange their inputs with ds_bpermute (no
//     LDS memory, no workgroup barrier).  Every lane evaluates the reference's rule for ITS output only -- out_i is
//     a fold over the other inputs in slot order in every rule (arithmetic.rs:214-246, 347-379, 487-521, 942-999),
//     so the per-lane folds perform exactly the operations the row-at-a-time evaluation of kernels.hip.h
//     (rule_check_node) performs for that output: bit-identical results, with the O(d^2) work of a row spread over d
//     lanes and the transcendental function of an input evaluated once, by its own lane;
//   * messages are stored in lane order ([chunk][lane]): every access of the check side is a coalesced segment and
//     only ever written by the lane that owns it; soft values are gathered / scattered by variable index.  Data that
//     crosses a barrier is written with plain (write-through) stores and read with nontemporal loads (latency.hip.h);
//   * LAYERED: the rows of a dependency level share no variable, so their in-place updates commute
//     (horizontal_layered.rs:105-110 processes rows 0..m in order; device_decoder.hip builds the levels): a level is
//     one parallel step, levels are separated by the XCD-local barrier of latency.hip.h; the syndrome of hard(Qv)
//     after every iteration (:66-78);
//   * FLOODING (flooding.rs:51-125): check-node phase over all rows -- x = L - c2v is the variable node's own
//     subtraction (arithmetic.rs:152), evaluated by the consumer as in the batched kernels -- with the row parities of
//     hard(L) of the previous iteration fused in (the convergence vote rides on the barrier); variable-node phase with
//     a lane per variable: the slot-ordered sum from -0.0 (arithmetic.rs:140-156).  Two barriers per iteration.
//
// Per-codeword semantics are those of the batch path: pre-check on the raw input (iterations 0), stop at the first
// zero syndrome, -1 after max_iterations with the last hard decisions, the max_iterations = 0 corners.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "kernels.hip.h"
#include "kernels_i8.hip.h"
#include "latency.hip.h"

namespace ldpc {
namespace dev {

struct EdgeLatTables {
  uint32_t n, m, n_levels, n_chunks;
  const uint32_t *level_chunk;  // [n_levels+1] first chunk of a level (flooding: one level, all rows)
  const uint32_t *lane_var;     // [n_chunks*64] variable of the lane's edge, kNoLane for padding
  const uint32_t *lane_info;    // [n_chunks*64] slot of the edge in its row | degree of the row << 8 | largest degree in the chunk << 16
  const uint32_t *var_ptr;      // flooding: [n+1] slots of variable v in var_lane
  const uint32_t *var_lane;     // flooding: [E] lane index (chunk * 64 + lane) of the variable's j-th edge, cols[v] order
  const int32_t *src_block;     // depuncture map or null
  uint32_t block_size;
};
enum : uint32_t { kNoLane = 0xFFFFFFFFu };

struct EdgeLatState {  // 8 XCDs x kEdgeBundle codeword slots carved from one allocation: soft | msg | chan | rawhard
  char *base;
  size_t slot_bytes, off_msg, off_chan, off_rawhard;
  uint32_t *flags;  // [8 XCDs][2][kEdgeBundle] convergence votes of the bundled codewords (zeroed before every launch)
};

template <typename T>
__device__ __forceinline__ T lat_ld(const T *p) {
  return __builtin_nontemporal_load(p);  // bypasses the CU's L1: served by the XCD's L2
}

// the lane's output: the rule's fold over the OTHER inputs of its row, in slot order.
//   x     this lane's input (Qv - R)
//   first lane index (in the wavefront) of the row's slot 0;  i, d: this lane's slot and the row's degree
//   dmax  largest degree among the wavefront's rows (wave-uniform trip count)
// Inactive lanes (padding) pass d = 0 and take part in the exchanges only.
template <int RULE, typename T>
__device__ __forceinline__ T rule_edge(T x, uint32_t first, uint32_t i, uint32_t d, uint32_t dmax) {
  const int src0 = static_cast<int>(first);
  if constexpr (RULE == kRuleTanh) {
    // arithmetic.rs:347-379
    const T c = Limits<T>::tanh_clamp;
    T h = T(0.5) * x;
    if (h < -c) h = -c;
    if (h > c) h = c;
    const T t = m_tanh_clamped(h);
    T product = T(1.0);
    for (uint32_t j = 0; j < dmax; j++) {
      const T tj = __shfl(t, src0 + static_cast<int>(j), 64);
      if (j < d && j != i) product *= tj;
    }
    return two_atanh(product);
  } else if constexpr (RULE == kRulePhi) {
    // arithmetic.rs:214-246
    const T p = phi_fn(m_abs(x));
    const uint32_t neg = x < T(0.0) ? 1u : 0u;
    T sum = T(0.0);
    uint32_t sign = 0;
    for (uint32_t j = 0; j < dmax; j++) {
      const T pj = __shfl(p, src0 + static_cast<int>(j), 64);
      const uint32_t nj = static_cast<uint32_t>(__shfl(static_cast<int>(neg), src0 + static_cast<int>(j), 64));
      if (j < d) {
        sum += pj;
        sign ^= nj;
      }
    }
    const T y = phi_fn(sum - p);
    const uint32_t s = neg ? (sign ^ 1u) : sign;
    return s == 0 ? y : -y;
  } else if constexpr (RULE == kRuleMinstarapprox || RULE == kRuleMinsum) {
    // arithmetic.rs:487-521 (Minsum: SURVEY.md Appendix A.6)
    uint32_t sign = 0;
    bool have = RULE == kRuleMinsum;
    T acc = RULE == kRuleMinsum ? Limits<T>::inf() : T(0.0);
    for (uint32_t j = 0; j < dmax; j++) {
      T v = __shfl(x, src0 + static_cast<int>(j), 64);
      if (j < d && j != i) {
        if (v < T(0.0)) sign ^= 1u;
        v = m_abs(v);
        if (!have) {
          acc = v;
          have = true;
        } else if constexpr (RULE == kRuleMinsum) {
          acc = m_min(v, acc);
        } else {
          acc = m_max(m_min(v, acc) - m_corr(m_abs(v - acc)), T(0.0));
        }
      }
    }
    return sign == 0 ? acc : -acc;
  } else {
    // Aminstar, arithmetic.rs:942-999: every lane of the row evaluates the row's quantities (argmin = FIRST minimum)
    uint32_t argmin = 0, sign = 0;
    T vmin = T(0.0), xmin = T(0.0);
    for (uint32_t j = 0; j < dmax; j++) {
      const T v = __shfl(x, src0 + static_cast<int>(j), 64);
      if (j < d) {
        if (v < T(0.0)) sign ^= 1u;
        const T a = m_abs(v);
        if (j == 0 || a < vmin) {
          vmin = a;
          xmin = v;
          argmin = j;
        }
      }
    }
    bool have = false;
    T delta = T(0.0);
    for (uint32_t j = 0; j < dmax; j++) {
      T v = __shfl(x, src0 + static_cast<int>(j), 64);
      if (j < d && j != argmin) {
        v = m_abs(v);
        if (!have) {
          delta = v;
          have = true;
        } else {
          delta = m_min(v, delta) - m_corr(m_abs(v - delta)) + m_corr(v + delta);
        }
      }
    }
    if (i == argmin) return ((sign != 0) != (xmin < T(0.0))) ? -delta : delta;
    delta = m_min(delta, vmin) - m_corr(m_abs(delta - vmin)) + m_corr(delta + vmin);
    return ((sign != 0) != (x < T(0.0))) ? -delta : delta;
  }
}

// The 8-bit arithmetics (Minstarapproxi8* / Aminstari8*, arithmetic.rs:603-848, 1003-1257) on this path: T = int32_t
// holds the i16 soft values and the i8 messages one per word (the values are those of the batched kernels of
// kernels_i8.hip.h: |soft| <= 127 (1 + degree), so no intermediate ever leaves 16 bits); RULE = kRuleEdgeI8 and the
// rule's options arrive at run time.
enum : int { kRuleEdgeI8 = 64 };

// the lane's output for the 8-bit rules: magnitudes fold in slot order from 255, an identity of both fold steps
// (kernels_i8.hip.h); the sign is the parity of the other inputs' signs, and a zero magnitude stays zero
__device__ __forceinline__ int rule_edge_i8(int x, uint32_t first, uint32_t i, uint32_t d, uint32_t dmax, I8Opts o) {
  const int src0 = static_cast<int>(first);
  const uint32_t neg = x < 0 ? 1u : 0u;
  const int w = (x < 0 ? -x : x) | static_cast<int>(neg << 8);  // magnitude | sign << 8: one exchange per input
  uint32_t sign = 0, mag;
  if (!o.aminstar) {
    // arithmetic.rs:722-753
    uint32_t acc = 255u;
    for (uint32_t j = 0; j < dmax; j++) {
      const uint32_t wj = static_cast<uint32_t>(__shfl(w, src0 + static_cast<int>(j), 64));
      if (j < d && j != i) {
        sign ^= wj >> 8;
        acc = i8_minstar(wj & 0xFFu, acc);
      }
    }
    mag = acc;
  } else {
    // arithmetic.rs:1134-1191: the row's first minimum is the minimum of (|x| << 16 | slot)
    uint32_t key = ~0u;
    for (uint32_t j = 0; j < dmax; j++) {
      const uint32_t wj = static_cast<uint32_t>(__shfl(w, src0 + static_cast<int>(j), 64));
      if (j < d) {
        sign ^= wj >> 8;
        key = min(key, ((wj & 0xFFu) << 16) | j);
      }
    }
    sign ^= neg;
    const uint32_t argmin = key & 0xFFFFu;
    uint32_t delta = 255u;
    for (uint32_t j = 0; j < dmax; j++) {
      const uint32_t wj = static_cast<uint32_t>(__shfl(w, src0 + static_cast<int>(j), 64));
      if (j < d && j != argmin) delta = i8_aminstar(wj & 0xFFu, delta);
    }
    mag = i == argmin ? delta : i8_aminstar(delta, key >> 16);
  }
  if (o.hardlimit) mag = mag >= 100u ? 127u : mag;
  return (sign & 1u) ? -static_cast<int>(mag) : static_cast<int>(mag);
}

// the variable node's message to a check: L - c2v (arithmetic.rs:152); 8-bit: clipped to +-127 (:648)
template <typename T>
__device__ __forceinline__ T edge_v2c(T l, T m) {
  if constexpr (std::is_integral<T>::value) return i8_clip(l - m);
  else return l - m;
}
// the caller's LLR in the decoder's arithmetic (arithmetic.rs:194-196; 8-bit: :690-699)
template <typename T, typename SrcT>
__device__ __forceinline__ T edge_quantize(SrcT raw) {
  if constexpr (std::is_integral<T>::value) return i8_quantize(static_cast<double>(raw));
  else return static_cast<T>(raw);
}

// parity of every row of the wavefront's chunk over `bit` (lane = edge): odd rows raise their first lane
__device__ __forceinline__ bool chunk_has_odd_row(bool bit, uint32_t lane, uint32_t i, uint32_t d) {
  const uint64_t b = __builtin_amdgcn_ballot_w64(bit);
  const uint64_t mask = d >= 64 ? ~0ull : ((1ull << d) - 1ull);
  return d != 0 && i == 0 && (__popcll((b >> lane) & mask) & 1u) != 0;
}

// BUNDLES: an XCD decodes up to `bundle` (<= kEdgeBundle) codewords of a call at once -- every phase walks the chunks
// of all of them, so a level's barrier is paid once per bundle, not once per codeword: 64 codewords take about the
// time of 8.  With one codeword per XCD the convergence vote rides on the barrier; with more, the wavefronts that
// saw an odd row write the vote's number into a flag per codeword (two flag sets, alternating: a set is rewritten
// only after every reader of its previous use has passed a later barrier).
enum : uint32_t { kEdgeBundle = 8 };

template <int RULE, typename T, typename SrcT, bool LAYERED>
__global__ __launch_bounds__(1024) void latency_edge_kernel(EdgeLatTables g, EdgeLatState slots, LatencySync *sync,
                                                            const SrcT *__restrict__ llrs, uint32_t input_len, uint32_t batch,
                                                            uint32_t max_iterations, uint8_t *__restrict__ bits,
                                                            uint32_t out_len, int32_t *__restrict__ iterations,
                                                            SrcT *__restrict__ posterior, uint32_t *error_word,
                                                            uint32_t bundle, I8Opts o) {
  constexpr bool I8 = RULE == kRuleEdgeI8;
  static_assert(I8 == std::is_integral<T>::value, "the 8-bit rules compute in int32_t words");
  __shared__ uint32_t s_slot, s_count, s_rank, s_nx;
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 0) {
    s_slot = __hip_atomic_fetch_add(&sync->arrived[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&sync->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // census: every workgroup of the grid is resident and has reported (the grid is sized to fit)
    bool dead = false;
    lat_spin_until(&sync->total, gridDim.x, error_word, &dead);
    uint32_t nx = 0, rank = 0;
    for (uint32_t x = 0; x < 8; x++) {
      const uint32_t a = lat_atomic_load(&sync->arrived[x]);
      if (a != 0) {
        if (x < xcc) rank++;
        nx++;
      }
    }
    s_count = lat_atomic_load(&sync->arrived[xcc]);
    s_rank = rank;
    s_nx = nx;
  }
  __syncthreads();
  const uint32_t count = s_count, nthreads = count * blockDim.x, t0 = s_slot * blockDim.x + threadIdx.x;
  const uint32_t nwaves = lat_uniform(nthreads >> 6), w0 = lat_uniform(t0 >> 6), lane = threadIdx.x & 63u;
  LatEpoch epoch;
  uint64_t (*const bar)[16] = sync->barrier[xcc];
  const uint32_t my_slot = s_slot;
  const uint32_t n = g.n, n_lanes = g.n_chunks * 64;
  const TablePtr level_chunk = table_ptr(g.level_chunk);
  uint32_t *const flags = slots.flags + s_rank * 2 * kEdgeBundle;  // [2][kEdgeBundle]: number of the last vote that saw an odd row
  uint32_t vote_no = 0;
  // state of codeword j of the bundle
  auto soft_of = [&](uint32_t j) { return reinterpret_cast<T *>(slots.base + (size_t(s_rank) * kEdgeBundle + j) * slots.slot_bytes); };
  auto msg_of = [&](uint32_t j) { return reinterpret_cast<T *>(reinterpret_cast<char *>(soft_of(j)) + slots.off_msg); };
  auto chan_of = [&](uint32_t j) { return reinterpret_cast<T *>(reinterpret_cast<char *>(soft_of(j)) + slots.off_chan); };
  auto raw_of = [&](uint32_t j) { return reinterpret_cast<uint8_t *>(soft_of(j)) + slots.off_rawhard; };

  // bundles of `bundle` consecutive codewords; the XCDs that have workgroups share them round-robin
  for (uint32_t cw0 = s_rank * bundle; cw0 < batch; cw0 += s_nx * bundle) {
    const uint32_t kb = min(bundle, batch - cw0);

    // barrier + which codewords of `candidates` had an odd row somewhere (oddmask: this thread's view, bit j)
    auto vote = [&](uint32_t oddmask, uint32_t candidates) -> uint32_t {
      if (kb == 1) return xcd_barrier(bar, count, my_slot, &epoch, error_word, oddmask & 1u) ? 1u : 0u;
      vote_no += 1;
      uint32_t *const f = flags + (vote_no & 1u) * kEdgeBundle;
      for (uint32_t j = 0; j < kb; j++)
        if (__builtin_amdgcn_ballot_w64(((oddmask >> j) & 1u) != 0) != 0 && lane == 0) f[j] = vote_no;  // same value from all
      xcd_barrier(bar, count, my_slot, &epoch, error_word);
      uint32_t m = 0;
      for (uint32_t j = 0; j < kb; j++)
        if (((candidates >> j) & 1u) && lat_atomic_load(f + j) == vote_no) m |= 1u << j;
      return m;
    };

    // ingest: depuncture (puncturing.rs:83-101), quantise (arithmetic.rs:194-196), raw hard decisions for the
    // pre-check; messages = +0.0: the first iteration's `x - 0.0` (and `out - 0.0`) is the reference's initial state
    for (uint32_t idx = t0; idx < kb * n; idx += nthreads) {
      const uint32_t j = idx / n, v = idx - j * n;
      const SrcT *src = llrs + size_t(cw0 + j) * input_len;
      SrcT raw;
      if (g.src_block) {
        const int32_t sb = g.src_block[v / g.block_size];
        raw = sb < 0 ? SrcT(0.0) : src[size_t(sb) * g.block_size + v % g.block_size];
      } else {
        raw = src[v];
      }
      const T quantized = edge_quantize<T, SrcT>(raw);
      soft_of(j)[v] = quantized;
      if (!LAYERED) chan_of(j)[v] = quantized;
      raw_of(j)[v] = raw <= SrcT(0.0) ? 1 : 0;
    }
    for (uint32_t idx = t0; idx < kb * n_lanes; idx += nthreads) {
      const uint32_t j = idx / n_lanes;
      msg_of(j)[idx - j * n_lanes] = T(0.0);
    }
    xcd_barrier(bar, count, my_slot, &epoch, error_word);

    // syndrome of hard decisions over every row of the codewords in `which`: the raw input (pre-check: flooding.rs:57-64,
    // horizontal_layered.rs:55-62) or the soft values (flooding.rs:69-79 at the last iteration, horizontal_layered.rs:66-78)
    auto odd_rows = [&](bool raw, uint32_t which) -> uint32_t {
      uint32_t oddmask = 0;
      for (uint32_t t = w0; t < g.n_chunks * kb; t += nwaves) {
        const uint32_t j = t / g.n_chunks, c = t - j * g.n_chunks;
        if (!((which >> j) & 1u)) continue;  // wave-uniform
        const uint32_t k = c * 64 + lane, var = g.lane_var[k], info = g.lane_info[k];
        const bool on = var != kNoLane;
        bool bit = false;
        if (on) bit = raw ? lat_load(raw_of(j) + var) != 0 : lat_ld(soft_of(j) + var) <= T(0.0);
        if (chunk_has_odd_row(bit, lane, info & 0xFFu, on ? ((info >> 8) & 0xFFu) : 0u)) oddmask |= 1u << j;
      }
      return oddmask;
    };

    int32_t result[kEdgeBundle];  // iterations on success, -1 while running / failed
#pragma unroll
    for (uint32_t j = 0; j < kEdgeBundle; j++) result[j] = -1;
    const uint32_t all = (1u << kb) - 1u;
    uint32_t live = vote(odd_rows(true, all), all);  // a codeword whose raw input has no odd row is done: 0 iterations
#pragma unroll
    for (uint32_t j = 0; j < kEdgeBundle; j++)
      if (j < kb && !((live >> j) & 1u)) result[j] = 0;

    if constexpr (LAYERED) {
      // One codeword: a wavefront's first chunk of a level is the same in every iteration; its table entries and its
      // R values (which only this lane ever writes) are requested one level ahead, before the barrier, so that a
      // level's critical path is the Qv gather, the rule, the stores and the barrier.
      uint32_t p_var = kNoLane, p_info = 0;
      T p_r = T(0.0);
      T *const msg0 = msg_of(0);
      auto prefetch = [&](uint32_t l) {
        const uint32_t c = level_chunk[l] + w0;
        p_var = kNoLane;
        p_info = 0;
        if (c < level_chunk[l + 1]) {
          const uint32_t k = c * 64 + lane;
          p_var = g.lane_var[k];
          p_info = g.lane_info[k];
          p_r = lat_ld(msg0 + k);  // (a padding lane's slot exists too: no dependence on the table entry)
        }
      };
      // (with a single level "one level ahead" would read this level's R before it is written)
      const bool ahead = kb == 1 && g.n_levels > 1;
      if (live && max_iterations > 0 && ahead) prefetch(0);
      for (uint32_t it = 1; live != 0 && it <= max_iterations; it++) {
        for (uint32_t l = 0; l < g.n_levels; l++) {
          const uint32_t lc0 = level_chunk[l], nch = level_chunk[l + 1] - lc0;
          const uint32_t next_level = l + 1 == g.n_levels ? 0 : l + 1;
          if (w0 >= nch && ahead) prefetch(next_level);
          for (uint32_t t = w0; t < nch * kb; t += nwaves) {
            const uint32_t j = t / nch, c = lc0 + (t - j * nch);
            if (!((live >> j) & 1u)) continue;  // wave-uniform
            T *__restrict__ soft = soft_of(j);
            T *__restrict__ msg = msg_of(j);
            const uint32_t k = c * 64 + lane;
            const bool pre = ahead && t == w0;
            uint32_t var, info;
            T r;
            if (pre) {
              var = p_var;
              info = p_info;
              r = p_r;
            } else {
              var = g.lane_var[k];
              info = g.lane_info[k];
              r = lat_ld(msg + k);
            }
            const bool on = var != kNoLane;
            const uint32_t i = info & 0xFFu, d = on ? ((info >> 8) & 0xFFu) : 0u;
            const uint32_t dmax = lat_uniform(info >> 16);  // largest degree in the chunk (every lane carries it)
            T q = T(0.0);
            if (on) q = lat_ld(soft + var);
            if (pre) prefetch(next_level);  // in flight behind the gather, consumed after the barrier
            const T x = edge_v2c(q, r);
            T out;
            if constexpr (I8) out = rule_edge_i8(x, lane - i, i, d, dmax, o);
            else out = rule_edge<RULE, T>(x, lane - i, i, d, dmax);
            if (on) {
              // Phi / Aminstar: Qv = x + out (arithmetic.rs:284-291, 1052-1065); the others: Qv += out - R (:423-424, 570-573;
              // 8-bit: :798, 1244-1254, x being the clipped copy of Qv - R)
              soft[var] = (RULE == kRulePhi || RULE == kRuleAminstar) ? (x + out) : (q + (out - r));
              msg[k] = out;
            }
          }
          xcd_barrier(bar, count, my_slot, &epoch, error_word);
        }
        const uint32_t still = vote(odd_rows(false, live), live);
#pragma unroll
        for (uint32_t j = 0; j < kEdgeBundle; j++)
          if (j < kb && ((live >> j) & 1u) && !((still >> j) & 1u)) result[j] = static_cast<int32_t>(it);
        live = still;
      }
    } else {
      // flooding: iteration `it` = check nodes from (L, c2v) of the previous one, with the parity of hard(L) over every
      // row fused in (it is the syndrome of iteration it - 1; iteration 1's is the pre-check above), then variable nodes
      uint32_t it = 1;
      for (; live != 0 && it <= max_iterations; it++) {
        uint32_t oddmask = it == 1 ? live : 0u;
        for (uint32_t t = w0; t < g.n_chunks * kb; t += nwaves) {
          const uint32_t j = t / g.n_chunks, c = t - j * g.n_chunks;
          if (!((live >> j) & 1u)) continue;  // wave-uniform
          T *__restrict__ msg = msg_of(j);
          const uint32_t k = c * 64 + lane, var = g.lane_var[k], info = g.lane_info[k];
          const bool on = var != kNoLane;
          const uint32_t i = info & 0xFFu, d = on ? ((info >> 8) & 0xFFu) : 0u;
          const uint32_t dmax = lat_uniform(info >> 16);
          T l = T(0.0), mo = T(0.0);
          if (on) {
            l = lat_ld(soft_of(j) + var);
            mo = lat_ld(msg + k);
          }
          if (chunk_has_odd_row(on && l <= T(0.0), lane, i, d)) oddmask |= 1u << j;
          const T x = edge_v2c(l, mo);  // v2c = L - c2v (arithmetic.rs:152)
          T out;
          if constexpr (I8) out = rule_edge_i8(x, lane - i, i, d, dmax, o);
          else out = rule_edge<RULE, T>(x, lane - i, i, d, dmax);
          if (on) msg[k] = out;
        }
        // (the messages just written are this iteration's; a codeword whose previous posterior turns out to be a
        // codeword simply does not use them)
        const uint32_t still = vote(oddmask, live);
#pragma unroll
        for (uint32_t j = 0; j < kEdgeBundle; j++)
          if (j < kb && ((live >> j) & 1u) && !((still >> j) & 1u)) result[j] = static_cast<int32_t>(it) - 1;
        live = still;
        if (live == 0) break;
        for (uint32_t idx = t0; idx < kb * n; idx += nthreads) {
          const uint32_t j = idx / n, v = idx - j * n;
          if (!((live >> j) & 1u)) continue;
          const T *__restrict__ msg = msg_of(j);
          const uint32_t e0 = g.var_ptr[v], e1 = g.var_ptr[v + 1];
          if constexpr (I8) {
            // arithmetic.rs:622-654: degree-one clipping of the input (:826-842), Jones clipping of the sum (:806-810)
            int llr = lat_ld(chan_of(j) + v);
            if (o.deg1clip && e1 - e0 == 1) llr = llr <= -116 ? -116 : (llr >= 116 ? 116 : llr);
            for (uint32_t e = e0; e < e1; e++) llr += lat_ld(msg + g.var_lane[e]);
            soft_of(j)[v] = o.jones ? i8_clip(llr) : llr;
          } else {
            T sum = -T(0.0);  // Rust's float Sum identity (arithmetic.rs:146)
            for (uint32_t e = e0; e < e1; e++) sum = sum + lat_ld(msg + g.var_lane[e]);
            soft_of(j)[v] = lat_ld(chan_of(j) + v) + sum;
          }
        }
        xcd_barrier(bar, count, my_slot, &epoch, error_word);
      }
      // the syndrome of the last posterior (flooding.rs:69-79 at iteration == max_iterations)
      if (live != 0 && max_iterations > 0) {
        const uint32_t still = vote(odd_rows(false, live), live);
#pragma unroll
        for (uint32_t j = 0; j < kEdgeBundle; j++)
          if (j < kb && ((live >> j) & 1u) && !((still >> j) & 1u)) result[j] = static_cast<int32_t>(max_iterations);
        live = still;
      }
    }

    // emit: converged at 0 -> the raw input's hard decisions; flooding with max_iterations = 0 and not a codeword ->
    // the reference's never-written output_llrs (all ones, 0.0; flooding.rs:27-28, 82-85); else hard(soft)
#pragma unroll
    for (uint32_t j = 0; j < kEdgeBundle; j++) {
      if (j < kb) {
        const int32_t res = result[j];
        const bool zero_fill = !LAYERED && res < 0 && max_iterations == 0;
        const T *__restrict__ soft = soft_of(j);
        const uint8_t *__restrict__ rawhard = raw_of(j);
        for (uint32_t v = t0; v < n; v += nthreads) {
          T val = lat_ld(soft + v);
          uint8_t bit = res == 0 ? static_cast<uint8_t>(lat_load(rawhard + v)) : (val <= T(0.0) ? 1 : 0);
          if (zero_fill) {
            val = T(0.0);
            bit = 1;
          }
          if (v < out_len) bits[size_t(cw0 + j) * out_len + v] = bit;
          // (8-bit: the soft output is the 8-bit LLR clip(llr), arithmetic.rs:651, 713-715)
          if constexpr (I8) {
            if (posterior) posterior[size_t(cw0 + j) * n + v] = static_cast<SrcT>(i8_clip(val));
          } else {
            if (posterior) posterior[size_t(cw0 + j) * n + v] = static_cast<SrcT>(val);
          }
        }
        if (t0 == 0 && iterations) iterations[cw0 + j] = res;
      }
    }
    xcd_barrier(bar, count, my_slot, &epoch, error_word);  // the slots' arrays are reused by this XCD's next bundle
  }
}

}  // namespace dev
}  // namespace ldpc
