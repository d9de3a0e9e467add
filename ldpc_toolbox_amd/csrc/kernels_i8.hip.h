// HIP kernels for the reference's 8-bit quantised arithmetics: Minstarapproxi8* and Aminstari8*
// (/root/reference/src/decoder/arithmetic.rs:582-897, 1074-1304).  Llr / messages are i8,
// the layered schedule's variable LLRs i16.  Integer arithmetic, so the results are exactly
// the reference's.
//
// Layout: the float path's tiled layout with narrower elements -- channel LLRs [N][tile] i8,
// posterior [N][tile] i16, messages [E][tile] i8.  A lane owns 4 consecutive codewords (one
// packed 32-bit word of four i8 values, or two words of four i16), a wave a 256-codeword slice.
// As on the float path v2c is not stored: the check-node kernel rebuilds
// clip(llr - m) (arithmetic.rs:648) from the stored i16 sum `llr` (after the optional Jones
// clipping) and the message.  The check row is staged in LDS as packed words, one column per
// thread, and evaluated in the reference's slot order.
#pragma once
#include "kernels.hip.h"

namespace ldpc {
namespace dev {

struct I8Opts {
  int aminstar;   // Aminstari8* (else Minstarapproxi8*)
  int jones;      // jones_clip!     arithmetic.rs:806-810
  int hardlimit;  // partial_hard_limit!  :812-824
  int deg1clip;   // degree_one_clipping! :826-842
};

// round(8 ln(1 + e^(-t/8))), t = 0.. while positive (arithmetic.rs:588-601); lookup beyond -> 0.
// The table is 6 6 5 5 4 4 3 3 3 3 2 2 2 2 1 (x9) 0...: the number of thresholds {1, 3, 5, 9, 13, 22} above t,
// i.e. the population count of a constant with bits 0, 2, 4, 8, 12, 21 shifted right by t.  Two vector
// instructions after the clamp of t, no memory: the lookup sits inside the serial fold of a check node and the
// kernels are bound by vector-ALU issue (DVB-S2 1/2 check-node launch, Minstarapproxi8: LDS byte table 1616 us,
// 64-bit shift of a packed constant 2008 us, six compare-and-add steps 2337 us; this form: see DESIGN.md 5).
constexpr uint32_t kI8TabBits = (1u << 0) | (1u << 2) | (1u << 4) | (1u << 8) | (1u << 12) | (1u << 21);
__device__ __forceinline__ uint32_t i8_tab(uint32_t t) {
  return static_cast<uint32_t>(__builtin_popcount(kI8TabBits >> min(t, 31u)));
}
__host__ __device__ constexpr int i8_table_entry(uint32_t t) {
  return int(t < 1) + int(t < 3) + int(t < 5) + int(t < 9) + int(t < 13) + int(t < 22);
}
constexpr bool i8_tab_matches_the_thresholds() {
  for (uint32_t t = 0; t < 1024; t++)
    if (__builtin_popcount(kI8TabBits >> (t < 31u ? t : 31u)) != i8_table_entry(t)) return false;
  return true;
}
static_assert(i8_tab_matches_the_thresholds(), "popcount form of the i8 lookup table");
// one fold step on magnitudes 0..255 held in 32-bit registers.  255 is an identity of both (min = the other
// operand, |difference| and sum >= 128 look up 0), so a fold may start from it instead of tracking "first".
// max(min(v, a) - lookup(|v - a|), 0)                                   arithmetic.rs:741
__device__ __forceinline__ uint32_t i8_minstar(uint32_t v, uint32_t a) {
  const uint32_t t = __builtin_amdgcn_sad_u8(v, a, 0u);  // |v - a|: both fit one byte
  return __builtin_elementwise_sub_sat(min(v, a), i8_tab(t));
}
// max(min(v, a) - lookup(|v - a|) + lookup(v saturating_add a), 0)       arithmetic.rs:1154-1156
// (the saturation at 127 only matters where the lookup is 0 anyway)
__device__ __forceinline__ uint32_t i8_aminstar(uint32_t v, uint32_t a) {
  const uint32_t t = __builtin_amdgcn_sad_u8(v, a, 0u);
  return __builtin_elementwise_sub_sat(min(v, a) + i8_tab(v + a), i8_tab(t));
}
// four i8 values in a word: |x| per byte (x >= -127), and -m per byte where s01's byte is 1 (m in 0..127;
// a zero stays zero: its negation would carry into the next byte)
__device__ __forceinline__ uint32_t pk_abs(uint32_t w) {
  const uint32_t s = (w >> 7) & 0x01010101u;
  return (w ^ (s * 255u)) + s;
}
__device__ __forceinline__ uint32_t pk_negate_where(uint32_t m, uint32_t s01) {
  s01 &= (m + 0x7F7F7F7Fu) >> 7;
  return (m ^ (s01 * 255u)) + s01;
}
__device__ __forceinline__ uint32_t ubyte_of(uint32_t w, int k) { return (w >> (8 * k)) & 0xFFu; }
__device__ __forceinline__ int i8_clip(int x) { return x >= 127 ? 127 : (x <= -127 ? -127 : x); }
__device__ __forceinline__ int i8_sat_add(int a, int b) {
  const int s = a + b;
  return s > 127 ? 127 : (s < -128 ? -128 : s);
}
__device__ __forceinline__ int i8_hardlimit(int x, int on) {
  if (!on) return x;
  return x <= -100 ? -127 : (x >= 100 ? 127 : x);
}
__device__ __forceinline__ int byte_of(uint32_t w, int k) { return static_cast<int8_t>(w >> (8 * k)); }
__device__ __forceinline__ uint32_t pack4(const int *v) {
  return (uint32_t(uint8_t(v[0]))) | (uint32_t(uint8_t(v[1])) << 8) | (uint32_t(uint8_t(v[2])) << 16) |
         (uint32_t(uint8_t(v[3])) << 24);
}
__device__ __forceinline__ int iabs(int x) { return x < 0 ? -x : x; }

// arithmetic.rs:690-699: clamp(round(8 x)) with round-half-away (Rust f64::round)
__device__ __forceinline__ int i8_quantize(double llr) {
  const double x = 8.0 * llr;
  if (x >= 127.0) return 127;
  if (x <= -127.0) return -127;
  if (x != x) return 0;  // Rust's float -> int `as` cast turns NaN into 0
  return static_cast<int>(round(x));
}

// Check node on the packed LDS column A[i*S] (four codewords per word), outputs to B[i*S]; A is left holding
// the magnitudes.  Minstarapprox: arithmetic.rs:722-753; A-Min*: :1134-1191 (min_by_key keeps the first minimum).
// The signs never enter the folds: the parity of the negative inputs is the XOR of the packed words (bit 7 of
// each byte), an output's sign is that parity without its own input's, applied to the packed magnitudes.
__device__ __forceinline__ void i8_check_node(uint32_t *A, uint32_t *B, uint32_t d, uint32_t S, I8Opts o) {
  uint32_t parity = 0;
  if (!o.aminstar) {
    for (uint32_t i = 0; i < d; i++) {
      const uint32_t w = A[i * S];
      B[i * S] = w;
      A[i * S] = pk_abs(w);
      parity ^= w;
    }
    // out_i folds the other magnitudes in slot order; the fold over the inputs before i is one running
    // prefix shared by every i (see rule_check_node in kernels.hip.h): same operations, same values
    uint32_t pacc[4] = {255u, 255u, 255u, 255u};
    for (uint32_t i = 0; i < d; i++) {
      uint32_t acc[4];
#pragma unroll
      for (int k = 0; k < 4; k++) acc[k] = pacc[k];
      for (uint32_t j = i + 1; j < d; j++) {
        const uint32_t mj = A[j * S];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = i8_minstar(ubyte_of(mj, k), acc[k]);
      }
      if (o.hardlimit) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = acc[k] >= 100u ? 127u : acc[k];
      }
      const uint32_t mag = acc[0] | (acc[1] << 8) | (acc[2] << 16) | (acc[3] << 24);
      B[i * S] = pk_negate_where(mag, ((parity ^ B[i * S]) >> 7) & 0x01010101u);
      const uint32_t mi = A[i * S];
#pragma unroll
      for (int k = 0; k < 4; k++) pacc[k] = i8_minstar(ubyte_of(mi, k), pacc[k]);
    }
    return;
  }
  // first minimum of |x|: the minimum of (|x| << 16 | slot)
  uint32_t key[4] = {~0u, ~0u, ~0u, ~0u};
  for (uint32_t i = 0; i < d; i++) {
    const uint32_t w = A[i * S];
    const uint32_t m = pk_abs(w);
    B[i * S] = w;
    A[i * S] = m;
    parity ^= w;
#pragma unroll
    for (int k = 0; k < 4; k++) key[k] = min(key[k], (ubyte_of(m, k) << 16) | i);
  }
  uint32_t delta[4] = {255u, 255u, 255u, 255u};
  for (uint32_t j = 0; j < d; j++) {
    const uint32_t m = A[j * S];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t next = i8_aminstar(ubyte_of(m, k), delta[k]);
      delta[k] = (key[k] & 0xFFFFu) == j ? delta[k] : next;
    }
  }
  uint32_t first[4], rest[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    first[k] = delta[k];
    rest[k] = i8_aminstar(delta[k], key[k] >> 16);
    if (o.hardlimit) {
      first[k] = first[k] >= 100u ? 127u : first[k];
      rest[k] = rest[k] >= 100u ? 127u : rest[k];
    }
  }
  for (uint32_t j = 0; j < d; j++) {
    uint32_t mag = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) mag |= ((key[k] & 0xFFFFu) == j ? first[k] : rest[k]) << (8 * k);
    B[j * S] = pk_negate_where(mag, ((parity ^ B[j * S]) >> 7) & 0x01010101u);
  }
}

// four i16 posterior values of a lane
struct alignas(8) Post4 {
  int16_t v[4];
};

// ---- ingest: caller's LLR rows -> chan (i8), post (i16 = chan), raw hard-decision ballots -------
template <typename SrcT>
__global__ __launch_bounds__(256) void ingest_i8_kernel(const SrcT *__restrict__ src, size_t src_stride,
                                                        uint32_t nb, uint32_t n, uint32_t G, uint32_t tile,
                                                        int8_t *__restrict__ chan, int16_t *__restrict__ post,
                                                        uint64_t *__restrict__ rawbits,
                                                        const int32_t *__restrict__ src_block,
                                                        uint32_t block_size) {
  __shared__ SrcT lds[64][65];
  const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
  const uint32_t v0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  for (uint32_t r = ty; r < 64; r += 4) {
    const uint32_t b = b0 + r, v = v0 + tx;
    SrcT val = SrcT(1.0);
    if (b < nb && v < n) {
      if (src_block) {
        const int32_t sb = src_block[v / block_size];
        val = sb < 0 ? SrcT(0.0) : src[size_t(b) * src_stride + size_t(sb) * block_size + v % block_size];
      } else {
        val = src[size_t(b) * src_stride + v];
      }
    }
    lds[r][tx] = val;
  }
  __syncthreads();
  const uint32_t W = G / 64;
  const size_t base = tile_base(b0, n, tile) + tx;
  for (uint32_t r = ty; r < 64; r += 4) {
    const uint32_t v = v0 + r;
    if (v < n) {
      const SrcT val = lds[tx][r];
      const int q = i8_quantize(static_cast<double>(val));
      chan[base + size_t(v) * tile] = static_cast<int8_t>(q);
      post[base + size_t(v) * tile] = static_cast<int16_t>(q);
      const uint64_t bal = __builtin_amdgcn_ballot_w64(val <= SrcT(0.0));
      if (tx == 0) rawbits[size_t(v) * W + blockIdx.y] = bal;
    }
  }
}

// ---- flooding check nodes ------------------------------------------------------------------------
// dynamic LDS: 2 * dmax * blockDim.x * 4 bytes + 32 (lookup table)
// (SCRATCH: rows beyond the LDS -- more than 320 edges -- keep the two columns in a per-wavefront region of HBM, as
// cn_staged_kernel does)
template <bool FIRST, bool SCRATCH = false>
__global__ void cn_i8_kernel(Graph g, Sched sc, State st, I8Opts o, const int8_t *__restrict__ chan,
                             const int16_t *__restrict__ post, int8_t *__restrict__ msg,
                             uint32_t *__restrict__ unsat_out, uint32_t dmax, uint32_t *__restrict__ scratch = nullptr) {
  constexpr int U = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr), edge_col = table_ptr(g.edge_col);
  const uint32_t tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t S = SCRATCH ? 64u : blockDim.x;
  uint32_t *A = SCRATCH ? scratch + size_t(wave) * 2u * dmax * 64u + lane : reinterpret_cast<uint32_t *>(smem) + threadIdx.x;
  uint32_t *B = A + size_t(dmax) * S;
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  chan += tile_base(b0, g.n_cols, sc) + lane * 4;
  post += tile_base(b0, g.n_cols, sc) + lane * 4;
  msg += tile_base(b0, g.n_edges, sc) + lane * 4;
  {
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < 4; k++) any_live = any_live || st.done[off + k] == 0;
    if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  }
  uint32_t odd_acc = 0;  // bit k: codeword k of this lane saw an odd check
  for (uint32_t c = node0; c < g.n_rows; c += sc.waves_per_chunk) {
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    const uint32_t d = e1 - e0;
    if (d == 0) continue;
    uint32_t par = 0;
    for (uint32_t i0 = 0; i0 < d; i0 += U) {
      Post4 pv[U];
      uint32_t cv[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < d) {
          const uint32_t v = edge_col[e0 + i0 + u];
          if (FIRST) {
            cv[u] = *reinterpret_cast<const uint32_t *>(chan + size_t(v) * tile);
          } else {
            pv[u] = *reinterpret_cast<const Post4 *>(post + size_t(v) * tile);
            mv[u] = *reinterpret_cast<const uint32_t *>(msg + size_t(e0 + i0 + u) * tile);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < d) {
          int x[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (FIRST) {
              x[k] = byte_of(cv[u], k);  // first variable messages are the channel LLRs (flooding.rs:94-99)
            } else {
              const int l = pv[u].v[k];
              x[k] = i8_clip(l - byte_of(mv[u], k));  // arithmetic.rs:648
              if (l <= 0) par ^= 1u << k;
            }
          }
          A[(i0 + u) * S] = pack4(x);
        }
      }
    }
    odd_acc |= par;
    i8_check_node(A, B, d, S, o);
    for (uint32_t i0 = 0; i0 < d; i0 += U) {
#pragma unroll
      for (int u = 0; u < U; u++)
        if (i0 + u < d) *reinterpret_cast<uint32_t *>(msg + size_t(e0 + i0 + u) * tile) = B[(i0 + u) * S];
    }
  }
  if (!FIRST) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (odd_acc & (1u << k)) unsat_out[off + k] = 1u;
  }
}

// ---- flooding variable nodes (arithmetic.rs:622-654) -------------------------------------------
// llr = deg1clip(input) + sum of messages (i16), optional Jones clip; stored as the i16 posterior.
#ifdef LDPC_I8_KERNELS_TU  // not a template: compiled in ONE translation unit (run_group_i8.hip)
__global__ __launch_bounds__(256) void vn_i8_kernel(Graph g, Sched sc, State st, I8Opts o,
                                                    const int8_t *__restrict__ chan,
                                                    const int8_t *__restrict__ msg, int16_t *__restrict__ post,
                                                    const uint32_t *__restrict__ unsat_in,
                                                    uint32_t *__restrict__ unsat_clear, int32_t latch_iteration) {
  constexpr int U = 8;
  uint32_t *__restrict__ n_active = st.n_active;
  if (*n_active == 0) return;
  const TablePtr col_ptr = table_ptr(g.col_ptr), col_edge = table_ptr(g.col_edge);
  const uint32_t tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, v_first;
  wave_slot(sc, wave, &chunk, &v_first);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  chan += tile_base(b0, g.n_cols, sc) + lane * 4;
  post += tile_base(b0, g.n_cols, sc) + lane * 4;
  msg += tile_base(b0, g.n_edges, sc) + lane * 4;
  bool skip[4];
  bool any_live = false, all = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool was_done = st.done[off + k] != 0;
    const bool converged = !was_done && unsat_in != nullptr && unsat_in[off + k] == 0;
    skip[k] = was_done || converged;
    any_live = any_live || !skip[k];
    all = all && !skip[k];
    if (v_first == 0) {
      if (converged) {
        st.done[off + k] = 1u;
        st.iters[off + k] = latch_iteration;
        atomicSub(n_active, 1u);
      }
      unsat_clear[off + k] = 0u;
    }
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  // index fetches of the next variable overlap the current variable's loads (as in vn_kernel):
  // without this a wave sits through three dependent latencies per variable (col_ptr -> col_edge
  // -> message rows) with only two or three 256-byte loads in flight
  const uint32_t n_cols = g.n_cols, waves_per_chunk = sc.waves_per_chunk;
  const uint32_t last_slot = g.n_edges ? g.n_edges - 1 : 0;
  uint32_t v = v_first, s0 = 0, s1 = 0, ed[U];
  if (v < n_cols) {
    s0 = col_ptr[v];
    s1 = col_ptr[v + 1];
  }
#pragma unroll
  for (int u = 0; u < U; u++) ed[u] = col_edge[min(s0 + u, last_slot)];
  while (v < n_cols) {
    const uint32_t ch = *reinterpret_cast<const uint32_t *>(chan + size_t(v) * tile);
    const uint32_t vn = v + waves_per_chunk;
    uint32_t ns0 = 0, ns1 = 0;
    if (vn < n_cols) {
      ns0 = col_ptr[vn];
      ns1 = col_ptr[vn + 1];
    }
    uint32_t ned[U];
    int llr[4] = {0, 0, 0, 0};
    for (uint32_t j0 = s0; j0 < s1; j0 += U) {
      uint32_t mv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (j0 + u < s1) {  // wave-uniform
          const uint32_t e = (j0 == s0) ? ed[u] : col_edge[j0 + u];
          mv[u] = *reinterpret_cast<const uint32_t *>(msg + size_t(e) * tile);
        }
      }
      if (j0 == s0) {
#pragma unroll
        for (int u = 0; u < U; u++) ned[u] = col_edge[min(ns0 + u, last_slot)];
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (j0 + u < s1) {
#pragma unroll
          for (int k = 0; k < 4; k++) llr[k] += byte_of(mv[u], k);
        }
    }
    if (s0 == s1) {
#pragma unroll
      for (int u = 0; u < U; u++) ned[u] = col_edge[min(ns0 + u, last_slot)];
    }
    Post4 out;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int in = byte_of(ch, k);
      if (o.deg1clip && s1 - s0 == 1) in = in <= -116 ? -116 : (in >= 116 ? 116 : in);
      const int l = in + llr[k];
      out.v[k] = static_cast<int16_t>(o.jones ? i8_clip(l) : l);
    }
    int16_t *dst = post + size_t(v) * tile;
    if (all) {
      *reinterpret_cast<Post4 *>(dst) = out;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (!skip[k]) dst[k] = out.v[k];
    }
    v = vn;
    s0 = ns0;
    s1 = ns1;
#pragma unroll
    for (int u = 0; u < U; u++) ed[u] = ned[u];
  }
}
#endif  // LDPC_I8_KERNELS_TU

// ---- layered schedule: one dependency level (arithmetic.rs:759-801, 1197-1257) -----------------
// dynamic LDS: 2 * dmax * blockDim.x * 4 bytes + 32 (lookup table)
template <bool FIRST, bool SCRATCH = false>
__global__ void hl_i8_kernel(Graph g, Sched sc, State st, I8Opts o, const uint32_t *__restrict__ level_rows,
                             uint32_t n_level_rows, int16_t *__restrict__ Q, int8_t *__restrict__ R,
                             uint32_t dmax, uint32_t *__restrict__ scratch = nullptr) {
  constexpr int U = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr), edge_col = table_ptr(g.edge_col);
  const uint32_t tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t S = SCRATCH ? 64u : blockDim.x;
  uint32_t *A = SCRATCH ? scratch + size_t(wave) * 2u * dmax * 64u + lane : reinterpret_cast<uint32_t *>(smem) + threadIdx.x;
  uint32_t *B = A + size_t(dmax) * S;
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  Q += tile_base(b0, g.n_cols, sc) + lane * 4;
  R += tile_base(b0, g.n_edges, sc) + lane * 4;
  bool frozen[4];
  bool any_live = false, all_live = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    frozen[k] = st.done[off + k] != 0;
    any_live = any_live || !frozen[k];
    all_live = all_live && !frozen[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  for (uint32_t idx = node0; idx < n_level_rows; idx += sc.waves_per_chunk) {
    const uint32_t c = table_ptr(level_rows)[idx];
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    const uint32_t d = e1 - e0;
    if (d == 0) continue;
    for (uint32_t i0 = 0; i0 < d; i0 += U) {
      Post4 qv[U];
      uint32_t rv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < d) {
          const uint32_t v = edge_col[e0 + i0 + u];
          qv[u] = *reinterpret_cast<const Post4 *>(Q + size_t(v) * tile);
          rv[u] = FIRST ? 0u : *reinterpret_cast<const uint32_t *>(R + size_t(e0 + i0 + u) * tile);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < d) {
          int x[4];
#pragma unroll
          for (int k = 0; k < 4; k++) x[k] = i8_clip(qv[u].v[k] - byte_of(rv[u], k));
          A[(i0 + u) * S] = pack4(x);
        }
      }
    }
    i8_check_node(A, B, d, S, o);
    for (uint32_t i = 0; i < d; i++) {
      const uint32_t v = edge_col[e0 + i];
      int16_t *qp = Q + size_t(v) * tile;
      int8_t *rp = R + size_t(e0 + i) * tile;
      const Post4 q = *reinterpret_cast<const Post4 *>(qp);
      const uint32_t r = FIRST ? 0u : *reinterpret_cast<const uint32_t *>(rp);
      const uint32_t ow = B[i * S];
      Post4 qn;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // Minstarapprox: Qv += out - R (:798); A-Min*: Qv = (Qv - R) + out (:1244-1254) -- the same
        // integer value
        qn.v[k] = static_cast<int16_t>(q.v[k] - byte_of(r, k) + byte_of(ow, k));
      }
      if (all_live) {
        *reinterpret_cast<Post4 *>(qp) = qn;
        *reinterpret_cast<uint32_t *>(rp) = ow;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (!frozen[k]) {
            qp[k] = qn.v[k];
            rp[k] = static_cast<int8_t>(byte_of(ow, k));
          }
      }
    }
  }
}

// hl_i8_kernel for levels whose rows have at most DMAX edges: Qv (four i16 per lane) and R (a packed
// word) of the whole row are loaded in one burst and kept in registers for the update -- no second
// pass over global memory (see hl_level_reg_kernel in kernels.hip.h).
template <int DMAX, bool FIRST>
__global__ void hl_i8_reg_kernel(Graph g, Sched sc, State st, I8Opts o, const uint32_t *__restrict__ level_rows,
                                 uint32_t n_level_rows, int16_t *__restrict__ Q, int8_t *__restrict__ R,
                                 uint32_t dmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr), edge_col = table_ptr(g.edge_col);
  const uint32_t S = blockDim.x, tile = sc.tile;
  uint32_t *A = reinterpret_cast<uint32_t *>(smem) + threadIdx.x;
  uint32_t *B = A + size_t(dmax) * S;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  Q += tile_base(b0, g.n_cols, sc) + lane * 4;
  R += tile_base(b0, g.n_edges, sc) + lane * 4;
  bool frozen[4];
  bool any_live = false, all_live = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    frozen[k] = st.done[off + k] != 0;
    any_live = any_live || !frozen[k];
    all_live = all_live && !frozen[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  for (uint32_t idx = node0; idx < n_level_rows; idx += sc.waves_per_chunk) {
    const uint32_t c = table_ptr(level_rows)[idx];
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    const uint32_t d = e1 - e0;
    if (d == 0) continue;
    uint32_t cols[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; i++) cols[i] = edge_col[e0 + min(uint32_t(i), d - 1)];
    Post4 q[DMAX];
    uint32_t r[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
        q[i] = *reinterpret_cast<const Post4 *>(Q + size_t(cols[i]) * tile);
        r[i] = FIRST ? 0u : *reinterpret_cast<const uint32_t *>(R + size_t(e0 + i) * tile);
      }
    }
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
        int x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = i8_clip(q[i].v[k] - byte_of(r[i], k));
        A[i * S] = pack4(x);
      }
    }
    i8_check_node(A, B, d, S, o);
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
        int16_t *qp = Q + size_t(cols[i]) * tile;
        int8_t *rp = R + size_t(e0 + i) * tile;
        const uint32_t ow = B[i * S];
        Post4 qn;
#pragma unroll
        for (int k = 0; k < 4; k++)
          qn.v[k] = static_cast<int16_t>(q[i].v[k] - byte_of(r[i], k) + byte_of(ow, k));
        if (all_live) {
          *reinterpret_cast<Post4 *>(qp) = qn;
          *reinterpret_cast<uint32_t *>(rp) = ow;
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (!frozen[k]) {
              qp[k] = qn.v[k];
              rp[k] = static_cast<int8_t>(byte_of(ow, k));
            }
        }
      }
    }
  }
}

}  // namespace dev
}  // namespace ldpc
