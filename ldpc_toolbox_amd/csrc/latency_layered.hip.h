// Small-batch (latency) path of the horizontal-layered schedule, all five float rules in f32: the reference's
// own call pattern is ONE codeword per decode call (/root/reference/src/c_api/decoder.rs:50-67,
// src/simulation/ber.rs:462-466), which through the batched kernels costs one launch per dependency level
// (32 per iteration on 5G NR BG1) with 1 lane in 64 useful.  Here, as in latency.hip.h:
//
//   * one persistent launch per call; a codeword is owned by one XCD (its Qv and R arrays stay in that XCD's L2),
//     up to 8 codewords of a call decode concurrently, more take turns;
//   * the rows of a dependency level share no variable, so their in-place updates commute
//     (horizontal_layered.rs:105-110 processes rows 0..m in order; device_decoder.hip builds the levels): a level
//     is one parallel step, levels are separated by the XCD-local barrier of latency.hip.h;
//   * a LANE owns one EDGE of one row.  The rows of a level are packed, whole, into wavefront-sized chunks of at
//     most 64 edge lanes; a row's d lanes sit next to each other in one wavefront and exchange their inputs with
//     ds_bpermute (no LDS memory, no workgroup barrier).  Every lane evaluates the reference's rule for ITS
//     output only -- out_i is a fold over the other inputs in slot order in every rule (arithmetic.rs:214-246,
//     347-379, 487-521, 942-999), so the per-lane folds perform exactly the operations the row-at-a-time
//     evaluation of kernels.hip.h (rule_check_node) performs for that output: bit-identical results, with the
//     O(d^2) work of a row spread over d lanes and the transcendental function of an input evaluated once
//     (by its own lane) instead of once per row pass;
//   * R is stored in lane order ([chunk][lane]): every access is a coalesced 256-byte segment and only ever
//     touched by the lane that owns it; Qv is gathered / scattered by variable index.  Data that crosses a barrier
//     is written with plain (write-through) stores and read with nontemporal loads (latency.hip.h).
//
// Per-codeword semantics are those of the batch path (horizontal_layered.rs:49-88): pre-check on the raw input
// (iterations 0), syndrome of hard(Qv) after every iteration, -1 after max_iterations with the last hard decisions.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hip.h"
#include "latency.hip.h"

namespace ldpc {
namespace dev {

struct LayeredLatTables {
  uint32_t n, m, n_levels, n_chunks;
  const uint32_t *level_chunk;  // [n_levels+1] first chunk of a level
  const uint32_t *lane_var;     // [n_chunks*64] variable of the lane's edge, kNoLane for padding
  const uint32_t *lane_info;    // [n_chunks*64] slot of the edge in its row | degree of the row << 8 | largest degree in the chunk << 16
  const int32_t *src_block;     // depuncture map or null
  uint32_t block_size;
};
enum : uint32_t { kNoLane = 0xFFFFFFFFu };

struct LayeredLatState {  // 8 codeword slots (one per XCD) carved from one allocation: Qv | R | rawhard
  char *base;
  size_t slot_bytes, off_r, off_rawhard;
};

// the lane's output: the rule's fold over the OTHER inputs of its row, in slot order.
//   x     this lane's input (Qv - R)
//   first lane index (in the wavefront) of the row's slot 0;  i, d: this lane's slot and the row's degree
//   dmax  largest degree among the wavefront's rows (wave-uniform trip count)
// Inactive lanes (padding) pass d = 0 and take part in the exchanges only.
template <int RULE>
__device__ __forceinline__ float rule_edge(float x, uint32_t first, uint32_t i, uint32_t d, uint32_t dmax) {
  const int src0 = static_cast<int>(first);
  if constexpr (RULE == kRuleTanh) {
    // arithmetic.rs:347-379
    const float c = Limits<float>::tanh_clamp;
    float h = 0.5f * x;
    if (h < -c) h = -c;
    if (h > c) h = c;
    const float t = m_tanh_clamped(h);
    float product = 1.0f;
    for (uint32_t j = 0; j < dmax; j++) {
      const float tj = __shfl(t, src0 + static_cast<int>(j), 64);
      if (j < d && j != i) product *= tj;
    }
    return 2.0f * atanh_rs(product);
  } else if constexpr (RULE == kRulePhi) {
    // arithmetic.rs:214-246
    const float p = phi_fn(m_abs(x));
    const uint32_t neg = x < 0.0f ? 1u : 0u;
    float sum = 0.0f;
    uint32_t sign = 0;
    for (uint32_t j = 0; j < dmax; j++) {
      const float pj = __shfl(p, src0 + static_cast<int>(j), 64);
      const uint32_t nj = static_cast<uint32_t>(__shfl(static_cast<int>(neg), src0 + static_cast<int>(j), 64));
      if (j < d) {
        sum += pj;
        sign ^= nj;
      }
    }
    const float y = phi_fn(sum - p);
    const uint32_t s = neg ? (sign ^ 1u) : sign;
    return s == 0 ? y : -y;
  } else if constexpr (RULE == kRuleMinstarapprox || RULE == kRuleMinsum) {
    // arithmetic.rs:487-521 (Minsum: SURVEY.md Appendix A.6)
    uint32_t sign = 0;
    bool have = RULE == kRuleMinsum;
    float acc = RULE == kRuleMinsum ? Limits<float>::inf() : 0.0f;
    for (uint32_t j = 0; j < dmax; j++) {
      float v = __shfl(x, src0 + static_cast<int>(j), 64);
      if (j < d && j != i) {
        if (v < 0.0f) sign ^= 1u;
        v = m_abs(v);
        if (!have) {
          acc = v;
          have = true;
        } else if constexpr (RULE == kRuleMinsum) {
          acc = m_min(v, acc);
        } else {
          acc = m_max(m_min(v, acc) - m_corr(m_abs(v - acc)), 0.0f);
        }
      }
    }
    return sign == 0 ? acc : -acc;
  } else {
    // Aminstar, arithmetic.rs:942-999: every lane of the row evaluates the row's quantities (argmin = FIRST minimum)
    uint32_t argmin = 0, sign = 0;
    float vmin = 0.0f, xmin = 0.0f;
    for (uint32_t j = 0; j < dmax; j++) {
      const float v = __shfl(x, src0 + static_cast<int>(j), 64);
      if (j < d) {
        if (v < 0.0f) sign ^= 1u;
        const float a = m_abs(v);
        if (j == 0 || a < vmin) {
          vmin = a;
          xmin = v;
          argmin = j;
        }
      }
    }
    bool have = false;
    float delta = 0.0f;
    for (uint32_t j = 0; j < dmax; j++) {
      float v = __shfl(x, src0 + static_cast<int>(j), 64);
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
    if (i == argmin) return ((sign != 0) != (xmin < 0.0f)) ? -delta : delta;
    delta = m_min(delta, vmin) - m_corr(m_abs(delta - vmin)) + m_corr(delta + vmin);
    return ((sign != 0) != (x < 0.0f)) ? -delta : delta;
  }
}

// parity of every row of the wavefront's chunk over `bit` (lane = edge): odd rows raise their first lane
__device__ __forceinline__ bool chunk_has_odd_row(bool bit, uint32_t lane, uint32_t i, uint32_t d) {
  const uint64_t b = __builtin_amdgcn_ballot_w64(bit);
  const uint64_t mask = d >= 64 ? ~0ull : ((1ull << d) - 1ull);
  return d != 0 && i == 0 && (__popcll((b >> lane) & mask) & 1u) != 0;
}

template <int RULE, typename SrcT>
__global__ __launch_bounds__(1024) void latency_layered_kernel(LayeredLatTables g, LayeredLatState slots, LatencySync *sync,
                                                               const SrcT *__restrict__ llrs, uint32_t input_len,
                                                               uint32_t batch, uint32_t max_iterations,
                                                               uint8_t *__restrict__ bits, uint32_t out_len,
                                                               int32_t *__restrict__ iterations,
                                                               SrcT *__restrict__ posterior, uint32_t *error_word) {
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
  const uint32_t n = g.n;
  const TablePtr level_chunk = table_ptr(g.level_chunk);

  // the XCDs that have workgroups share the codewords round-robin
  for (uint32_t cw = s_rank; cw < batch; cw += s_nx) {
    char *const slot = slots.base + size_t(s_rank) * slots.slot_bytes;
    float *__restrict__ qv = reinterpret_cast<float *>(slot);
    float *__restrict__ rr = reinterpret_cast<float *>(slot + slots.off_r);
    uint8_t *__restrict__ rawhard = reinterpret_cast<uint8_t *>(slot + slots.off_rawhard);
    const SrcT *src = llrs + size_t(cw) * input_len;

    // ingest: depuncture (puncturing.rs:83-101), quantise (`x as f32`), raw hard decisions for the pre-check;
    // R = +0.0: the first iteration's `Qv - 0.0` and `out - 0.0` are the reference's initial state exactly
    for (uint32_t v = t0; v < n; v += nthreads) {
      SrcT raw;
      if (g.src_block) {
        const int32_t sb = g.src_block[v / g.block_size];
        raw = sb < 0 ? SrcT(0.0) : src[size_t(sb) * g.block_size + v % g.block_size];
      } else {
        raw = src[v];
      }
      qv[v] = static_cast<float>(raw);
      rawhard[v] = raw <= SrcT(0.0) ? 1 : 0;
    }
    for (uint32_t k = t0; k < g.n_chunks * 64; k += nthreads) rr[k] = 0.0f;
    xcd_barrier(bar, count, my_slot, &epoch, error_word);

    // syndrome of hard decisions over every row: raw input (pre-check, horizontal_layered.rs:55-62) or Qv (:66-78)
    auto any_odd_row = [&](bool raw) {
      bool odd = false;
      for (uint32_t c = w0; c < g.n_chunks; c += nwaves) {
        const uint32_t k = c * 64 + lane, var = g.lane_var[k], info = g.lane_info[k];
        const bool on = var != kNoLane;
        bool bit = false;
        if (on) bit = raw ? lat_load(rawhard + var) != 0 : lat_load(qv + var) <= 0.0f;
        odd = odd || chunk_has_odd_row(bit, lane, info & 0xFFu, on ? ((info >> 8) & 0xFFu) : 0u);
      }
      return odd;
    };

    int32_t result = -1;  // iterations on success
    if (!xcd_barrier(bar, count, my_slot, &epoch, error_word, any_odd_row(true) ? 1u : 0u)) result = 0;
    // A wavefront's first chunk of a level is the same in every iteration: its table entries and its R values
    // (which only this lane ever writes) are requested one level ahead, before the barrier, so that a level's
    // critical path is the Qv gather, the rule, the stores and the barrier.
    uint32_t p_var = kNoLane, p_info = 0;
    float p_r = 0.0f;
    auto prefetch = [&](uint32_t l) {
      const uint32_t c = level_chunk[l] + w0;
      p_var = kNoLane;
      p_info = 0;
      if (c < level_chunk[l + 1]) {
        const uint32_t k = c * 64 + lane;
        p_var = g.lane_var[k];
        p_info = g.lane_info[k];
        p_r = lat_load(rr + k);  // (a padding lane's slot exists too: no dependence on the table entry)
      }
    };
    const bool ahead = g.n_levels > 1;  // (with a single level "one level ahead" would read this level's R before it is written)
    if (result < 0 && max_iterations > 0 && ahead) prefetch(0);
    for (uint32_t it = 1; result < 0 && it <= max_iterations; it++) {
      for (uint32_t l = 0; l < g.n_levels; l++) {
        const uint32_t c0 = level_chunk[l] + w0, c1 = level_chunk[l + 1];
        const uint32_t next_level = l + 1 == g.n_levels ? 0 : l + 1;
        if (c0 >= c1 && ahead) prefetch(next_level);
        for (uint32_t c = c0; c < c1; c += nwaves) {
          const uint32_t k = c * 64 + lane;
          uint32_t var, info;
          float r;
          if (c == c0 && ahead) {
            var = p_var;
            info = p_info;
            r = p_r;
          } else {
            var = g.lane_var[k];
            info = g.lane_info[k];
            r = lat_load(rr + k);
          }
          const bool on = var != kNoLane;
          const uint32_t i = info & 0xFFu, d = on ? ((info >> 8) & 0xFFu) : 0u;
          const uint32_t dmax = lat_uniform(info >> 16);  // largest degree in the chunk (every lane carries it)
          float q = 0.0f;
          if (on) q = lat_load(qv + var);
          if (c == c0 && ahead) prefetch(next_level);  // in flight behind the gather, consumed after the barrier
          const float x = q - r;
          const float out = rule_edge<RULE>(x, lane - i, i, d, dmax);
          if (on) {
            // Phi / Aminstar: Qv = x + out (arithmetic.rs:284-291, 1052-1065); the others: Qv += out - R (:423-424, 570-573)
            qv[var] = (RULE == kRulePhi || RULE == kRuleAminstar) ? (x + out) : (q + (out - r));
            rr[k] = out;
          }
        }
        xcd_barrier(bar, count, my_slot, &epoch, error_word);
      }
      if (!xcd_barrier(bar, count, my_slot, &epoch, error_word, any_odd_row(false) ? 1u : 0u)) result = static_cast<int32_t>(it);
    }

    // emit: converged at 0 -> the raw input's hard decisions; otherwise hard(Qv); the soft output is Qv
    for (uint32_t v = t0; v < n; v += nthreads) {
      const float val = lat_load(qv + v);
      const uint8_t bit = result == 0 ? static_cast<uint8_t>(lat_load(rawhard + v)) : (val <= 0.0f ? 1 : 0);
      if (v < out_len) bits[size_t(cw) * out_len + v] = bit;
      if (posterior) posterior[size_t(cw) * n + v] = static_cast<SrcT>(val);
    }
    if (t0 == 0 && iterations) iterations[cw] = result;
    xcd_barrier(bar, count, my_slot, &epoch, error_word);  // the slot's arrays are reused by this XCD's next codeword
  }
}

}  // namespace dev
}  // namespace ldpc
