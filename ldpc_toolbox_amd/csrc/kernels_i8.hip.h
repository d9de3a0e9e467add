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
// The table lives in LDS (32 bytes behind the staged columns): a lookup sits inside the serial
// fold of a check node, where a constant-memory load would put a global-memory latency on every
// step; 22 bytes span six banks, so any mix of indices is conflict-free.  (Arithmetic forms were
// timed against it on DVB-S2 1/2, check-node launch: LDS table 1616 us, a 63-bit packed constant
// with a 64-bit shift 2008 us, six compare-and-add steps 2337 us: the kernel is VALU-bound.)
__device__ __forceinline__ int i8_table_entry(uint32_t t) {
  return int(t < 1) + int(t < 3) + int(t < 5) + int(t < 9) + int(t < 13) + int(t < 22);
}
__device__ __forceinline__ void i8_table_init(uint8_t *tab) {
  if (threadIdx.x < 32) tab[threadIdx.x] = static_cast<uint8_t>(i8_table_entry(threadIdx.x));
  __syncthreads();
}
__device__ __forceinline__ int i8_lookup(const uint8_t *tab, int t) {
  return tab[min(static_cast<uint32_t>(t), 22u)];  // negative t wraps to a large index -> 0
}
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

// Check node on the packed LDS column A[i*S] (four codewords per word), outputs to B[i*S].
// Minstarapprox: arithmetic.rs:722-753; A-Min*: :1134-1191 (min_by_key keeps the first minimum).
__device__ __forceinline__ void i8_check_node(const uint32_t *A, uint32_t *B, uint32_t d, uint32_t S, I8Opts o,
                                              const uint8_t *tab) {
  if (!o.aminstar) {
    // shared running prefix of the fold (see rule_check_node in kernels.hip.h): same operations
    uint32_t psign = 0, phave = 0;  // bit k: codeword k
    int pacc[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < d; i++) {
      uint32_t sign = psign, have = phave;
      int acc[4];
#pragma unroll
      for (int k = 0; k < 4; k++) acc[k] = pacc[k];
      for (uint32_t j = i + 1; j < d; j++) {
        const uint32_t wj = A[j * S];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int v = byte_of(wj, k);
          if (v < 0) sign ^= 1u << k;
          v = iabs(v);
          if (!(have & (1u << k))) {
            acc[k] = v;
          } else {
            const int m = min(v, acc[k]) - i8_lookup(tab, iabs(v - acc[k]));
            acc[k] = m > 0 ? m : 0;
          }
        }
        have = 0xFu;
      }
      int outv[4];
      const uint32_t wi = A[i * S];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        outv[k] = i8_hardlimit((sign & (1u << k)) == 0 ? acc[k] : -acc[k], o.hardlimit);
        int v = byte_of(wi, k);
        if (v < 0) psign ^= 1u << k;
        v = iabs(v);
        if (!phave) {
          pacc[k] = v;
        } else {
          const int m = min(v, pacc[k]) - i8_lookup(tab, iabs(v - pacc[k]));
          pacc[k] = m > 0 ? m : 0;
        }
      }
      phave = 0xFu;
      B[i * S] = pack4(outv);
    }
    return;
  }
  int argmin[4], vmin[4], delta[4];
  uint32_t sign[4];
  bool have[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    argmin[k] = 0;
    vmin[k] = iabs(byte_of(A[0], k));
    delta[k] = 0;
    sign[k] = 0;
    have[k] = false;
  }
  for (uint32_t i = 1; i < d; i++) {
    const uint32_t w = A[i * S];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int a = iabs(byte_of(w, k));
      if (a < vmin[k]) {
        vmin[k] = a;
        argmin[k] = static_cast<int>(i);
      }
    }
  }
  for (uint32_t j = 0; j < d; j++) {
    const uint32_t w = A[j * S];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int v = byte_of(w, k);
      if (v < 0) sign[k] ^= 1u;
      if (static_cast<int>(j) != argmin[k]) {
        v = iabs(v);
        if (!have[k]) {
          delta[k] = v;
          have[k] = true;
        } else {
          const int m = min(v, delta[k]) - i8_lookup(tab, iabs(v - delta[k])) + i8_lookup(tab, i8_sat_add(v, delta[k]));
          delta[k] = m > 0 ? m : 0;
        }
      }
    }
  }
  int first_hl[4], rest_hl[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    first_hl[k] = i8_hardlimit(delta[k], o.hardlimit);
    const int m = min(delta[k], vmin[k]) - i8_lookup(tab, iabs(delta[k] - vmin[k])) + i8_lookup(tab, i8_sat_add(delta[k], vmin[k]));
    rest_hl[k] = i8_hardlimit(m > 0 ? m : 0, o.hardlimit);
  }
  for (uint32_t j = 0; j < d; j++) {
    const uint32_t w = A[j * S];
    int outv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int v = byte_of(w, k);
      const int mag = (static_cast<int>(j) == argmin[k]) ? first_hl[k] : rest_hl[k];
      outv[k] = ((sign[k] != 0) != (v < 0)) ? -mag : mag;
    }
    B[j * S] = pack4(outv);
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
template <bool FIRST>
__global__ void cn_i8_kernel(Graph g, Sched sc, State st, I8Opts o, const int8_t *__restrict__ chan,
                             const int16_t *__restrict__ post, int8_t *__restrict__ msg,
                             uint32_t *__restrict__ unsat_out, uint32_t dmax) {
  constexpr int U = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const uint32_t *__restrict__ row_ptr = g.row_ptr;
  const uint32_t *__restrict__ edge_col = g.edge_col;
  const uint32_t S = blockDim.x, tile = sc.tile;
  uint32_t *A = reinterpret_cast<uint32_t *>(smem) + threadIdx.x;
  uint32_t *B = A + size_t(dmax) * S;
  uint8_t *tab = smem + size_t(2) * dmax * S * 4;
  i8_table_init(tab);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  chan += tile_base(b0, g.n_cols, tile) + lane * 4;
  post += tile_base(b0, g.n_cols, tile) + lane * 4;
  msg += tile_base(b0, g.n_edges, tile) + lane * 4;
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
    i8_check_node(A, B, d, S, o, tab);
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
__global__ __launch_bounds__(256) void vn_i8_kernel(Graph g, Sched sc, State st, I8Opts o,
                                                    const int8_t *__restrict__ chan,
                                                    const int8_t *__restrict__ msg, int16_t *__restrict__ post,
                                                    const uint32_t *__restrict__ unsat_in,
                                                    uint32_t *__restrict__ unsat_clear, int32_t latch_iteration) {
  constexpr int U = 8;
  uint32_t *__restrict__ n_active = st.n_active;
  if (*n_active == 0) return;
  const uint32_t *__restrict__ col_ptr = g.col_ptr;
  const uint32_t *__restrict__ col_edge = g.col_edge;
  const uint32_t tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, v_first;
  wave_slot(sc, wave, &chunk, &v_first);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  chan += tile_base(b0, g.n_cols, tile) + lane * 4;
  post += tile_base(b0, g.n_cols, tile) + lane * 4;
  msg += tile_base(b0, g.n_edges, tile) + lane * 4;
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

// ---- layered schedule: one dependency level (arithmetic.rs:759-801, 1197-1257) -----------------
// dynamic LDS: 2 * dmax * blockDim.x * 4 bytes + 32 (lookup table)
template <bool FIRST>
__global__ void hl_i8_kernel(Graph g, Sched sc, State st, I8Opts o, const uint32_t *__restrict__ level_rows,
                             uint32_t n_level_rows, int16_t *__restrict__ Q, int8_t *__restrict__ R,
                             uint32_t dmax) {
  constexpr int U = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const uint32_t *__restrict__ row_ptr = g.row_ptr;
  const uint32_t *__restrict__ edge_col = g.edge_col;
  const uint32_t S = blockDim.x, tile = sc.tile;
  uint32_t *A = reinterpret_cast<uint32_t *>(smem) + threadIdx.x;
  uint32_t *B = A + size_t(dmax) * S;
  uint8_t *tab = smem + size_t(2) * dmax * S * 4;
  i8_table_init(tab);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  Q += tile_base(b0, g.n_cols, tile) + lane * 4;
  R += tile_base(b0, g.n_edges, tile) + lane * 4;
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
    const uint32_t c = level_rows[idx];
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
    i8_check_node(A, B, d, S, o, tab);
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
  const uint32_t *__restrict__ row_ptr = g.row_ptr;
  const uint32_t *__restrict__ edge_col = g.edge_col;
  const uint32_t S = blockDim.x, tile = sc.tile;
  uint32_t *A = reinterpret_cast<uint32_t *>(smem) + threadIdx.x;
  uint32_t *B = A + size_t(dmax) * S;
  uint8_t *tab = smem + size_t(2) * dmax * S * 4;
  i8_table_init(tab);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 256;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * 4;
  Q += tile_base(b0, g.n_cols, tile) + lane * 4;
  R += tile_base(b0, g.n_edges, tile) + lane * 4;
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
    const uint32_t c = level_rows[idx];
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
    i8_check_node(A, B, d, S, o, tab);
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
