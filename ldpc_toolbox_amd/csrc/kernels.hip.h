// HIP kernels of the batched BP decoder (gfx950 / CDNA4, wave64).
//
// Mapping used by every kernel: a LANE owns one codeword (VEC consecutive codewords in
// the streaming kernels), a WAVEFRONT owns one graph node (check row / variable) for a
// tile of 64*VEC codewords, so that
//   * all graph indices are wave-uniform (scalar loads, SGPR address math),
//   * every global access is a contiguous 256 B .. 1 KiB row segment,
//   * the per-node reductions (min1/min2/argmin/sign for min-sum, the slot-ordered
//     variable sum) run in registers with no cross-lane traffic.
// Cross-lane primitives are used where data really crosses codewords: ballots for the
// packed hard decisions and the "all codewords of my tile are finished" early-outs.
// The sum-product family stages the check row's edges in LDS ([slot][thread] columns,
// conflict-free) because those rules need random access to all d inputs.
//
// Arithmetic follows the reference rule by rule (citations at each function); compiled
// with -ffp-contract=off, comparisons instead of sign-bit tricks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "exact_math.h"
#include "fast_div.h"

// Timing-experiment switches (skip stores / gathers: WRONG results) exist only in builds made with
// EXTRA_HIPFLAGS=-DLDPC_EXPERIMENTS (tools/records_ab.py, tools/latency_probe.py); the product carries neither the
// kernel parameter nor the tests on it.
#ifdef LDPC_EXPERIMENTS
#define LDPC_DBG_PARAM(name) , uint32_t name
#define LDPC_DBG_ARG(x) , x
#else
#define LDPC_DBG_PARAM(name)
#define LDPC_DBG_ARG(x)
#endif

namespace ldpc {
namespace dev {

enum : int { kRulePhi = 0, kRuleTanh = 1, kRuleMinstarapprox = 2, kRuleAminstar = 3, kRuleMinsum = 4,
             kRuleTanhFast = 5, kRulePhiFast = 6 };  // "@fast": native exp2 / log2 / rcp, not bit-identical (f32 only)

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
  T v[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack(const T *p) {
  return *reinterpret_cast<const Pack<T, VEC> *>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_pack(T *p, const Pack<T, VEC> &x) {
  *reinterpret_cast<Pack<T, VEC> *>(p) = x;
}

// streamed-once data (messages): nontemporal accesses keep them from displacing the posterior
// rows that the check-node kernel re-reads out of L2 / Infinity Cache
template <typename T, int VEC>
struct VecOf {
  typedef T type __attribute__((ext_vector_type(VEC)));
};
template <typename T>
struct VecOf<T, 1> {
  typedef T type;
};
template <typename T, int VEC, bool NT>
__device__ __forceinline__ Pack<T, VEC> load_msg(const T *p) {
  if constexpr (NT) {
    using V = typename VecOf<T, VEC>::type;
    const V v = __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
    return __builtin_bit_cast(Pack<T, VEC>, v);
  } else {
    return load_pack<T, VEC>(p);
  }
}
template <typename T, int VEC, bool NT>
__device__ __forceinline__ void store_msg(T *p, const Pack<T, VEC> &x) {
  if constexpr (NT) {
    using V = typename VecOf<T, VEC>::type;
    __builtin_nontemporal_store(__builtin_bit_cast(V, x), reinterpret_cast<V *>(p));
  } else {
    store_pack<T, VEC>(p, x);
  }
}

__device__ __forceinline__ uint32_t uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// Graph tables are never written by a kernel.  Read through the constant address space a load whose index is
// wave-uniform is a scalar load (s_load_dword into an SGPR: no vector-memory instruction, no readfirstlane, and it
// does not share the in-order vmcnt counter with the data loads -- through a generic pointer the compiler has to
// assume the kernel's own stores may alias the tables and issues one vector load per index, which chains
// "index, wait, data, wait" edge after edge).
typedef const uint32_t __attribute__((address_space(4))) *TablePtr;
__device__ __forceinline__ TablePtr table_ptr(const uint32_t *p) { return (TablePtr)p; }

// Buffer addressing for the [row][tile] arrays: the descriptor of a wavefront's slice and the row offset
// (graph indices are wave-uniform) live in SGPRs, the lane's byte offset inside a row is one constant VGPR:
// a row access costs no vector address arithmetic (two 64-bit vector adds per access otherwise -- they count,
// the sum-product kernels are bound by vector-ALU issue).  NT: nontemporal, as load_msg / store_msg.
struct RowBuf {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ RowBuf row_buf(const void *p, uint64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  // (readfirstlane returns int: widen through uint32_t, or a low word with bit 31 set sign-extends)
  const uint64_t u = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(a >> 32)))) << 32) |
                     uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(a))));
  const uint32_t n = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(bytes);
  return RowBuf{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(u), 0, static_cast<int>(n), 0x00020000)};
}
template <typename T, bool NT>
__device__ __forceinline__ T row_load(const RowBuf &b, uint32_t lane_off, uint32_t row_off) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(b.r, lane_off, row_off, NT ? 2 : 0));
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(b.r, lane_off, row_off, NT ? 2 : 0));
  }
}
template <typename T, bool NT>
__device__ __forceinline__ void row_store(const RowBuf &b, uint32_t lane_off, uint32_t row_off, T v) {
  if constexpr (sizeof(T) == 4) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), b.r, lane_off, row_off, NT ? 2 : 0);
  } else {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), b.r, lane_off, row_off, NT ? 2 : 0);
  }
}

// Tiled codeword layout: an array of `rows` rows for G codewords is stored as
// [G / tile][rows][tile]; element (row r, codeword b) sits at
// tile_base(b - b % 64..., rows, tile) + r * tile + (offset of b inside its slice).
// A tile (default 256 codewords) is a self-contained sub-batch: its posterior array
// (N * tile * 4 B = 66 MB for DVB-S2) fits the 256 MB Infinity Cache, so the check-node
// kernel's d_v-fold re-reads of a posterior row are served on-die, and the set of pages a
// launch touches at any moment is small.  Waves are ordered tile-major.
__device__ __forceinline__ size_t tile_base(uint32_t b0, uint32_t rows, uint32_t tile) {
  return (size_t(b0 / tile) * rows) * tile + (b0 % tile);
}

// graph tables in HBM (shared by the whole batch) and the wave -> (tile slice, node) schedule
struct Graph {
  const uint32_t *row_ptr, *edge_col;  // checks: edge range, variable of each edge (rows[c] order)
  const uint32_t *col_ptr, *col_edge;  // variables: slot range, row-major edge id per slot (cols[v] order)
  uint32_t n_rows, n_cols, n_edges;
  // optional variable subset for vn_kernel (compacted CSC): item i is variable list_var[i] with
  // slots list_ptr[i]..list_ptr[i+1] of list_edge
  const uint32_t *list_var, *list_ptr, *list_edge;
  uint32_t n_list;
  // "L-free" variables (degree 1 or 2): their posterior is rebuilt by the check-node kernel from
  // the channel LLR and the two messages, so the variable-node kernel skips them.
  // edge_aux[e]: kAuxNone, or for an edge whose variable is L-free: the edge id of the variable's
  // other edge (kAuxSingle for degree 1), with kAuxWriter set on the variable's first slot
  const uint32_t *edge_aux;
  // row-record kernels (cn_minsum_rec_kernel): per-edge word, see kPeerKeep
  const uint32_t *edge_peer;
};
enum : uint32_t { kAuxNone = 0xFFFFFFFFu, kAuxWriter = 0x80000000u, kAuxSingle = 0x7FFFFFFEu, kAuxMask = 0x7FFFFFFFu };
struct Sched {
  uint32_t tile;             // codewords per layout tile
  uint32_t nchunks;          // wave-sized codeword slices in the group
  uint32_t waves_per_chunk;  // waves sharing one slice (node stride of a wave's loop)
  uint32_t slices_per_tile;  // wave order: tile, then node, then slice inside the tile
  uint32_t reverse;          // 1: the tiles are walked last to first (a launch that consumes what the previous launch
                             // produced tile by tile starts with the tiles it wrote last: those are still in the Infinity Cache)
  // the same numbers as multipliers (make_tiling fills them): waves per tile, slices per tile, codewords per tile
  FastDiv per_tile_div, spt_div, tile_div;
  uint32_t n_tiles;          // tiles covered by nchunks (the `reverse` order needs it)
};
__device__ __forceinline__ size_t tile_base(uint32_t b0, uint32_t rows, const Sched &sc) {
  const uint32_t t = fdiv_q(b0, sc.tile_div);
  return (size_t(t) * rows) * sc.tile + (b0 - t * sc.tile);
}
__device__ __forceinline__ uint32_t in_tile_of(uint32_t b0, const Sched &sc) { return b0 - fdiv_q(b0, sc.tile_div) * sc.tile; }

// wave -> (codeword slice, first node): tile-major, slices of one tile adjacent so that the waves
// of a workgroup read neighbouring segments of the same rows
__device__ __forceinline__ void wave_slot(const Sched &sc, uint32_t wave, uint32_t *chunk, uint32_t *node0) {
  uint32_t t = fdiv_q(wave, sc.per_tile_div);
  const uint32_t rem = wave - t * sc.per_tile_div.d;
  if (sc.reverse) t = t < sc.n_tiles ? sc.n_tiles - 1 - t : t;
  const uint32_t node = fdiv_q(rem, sc.spt_div);
  *chunk = t * sc.slices_per_tile + (rem - node * sc.slices_per_tile);
  *node0 = node;
}
// per-codeword decoder state of a group
struct State {
  uint32_t *done;      // 1 = finished (converged earlier, or padding beyond the batch)
  int32_t *iters;      // iteration at which it converged, -1 while running / failed
  uint32_t *n_active;  // codewords still running: every kernel returns at once when 0
  // Batch compaction (compact_* kernels): the group's live codewords occupy slots
  // [0, *n_slots) (a multiple of 256); slot_cw[s] = index of that codeword in the caller's
  // batch rows (kNoCodeword for padding).  Waves beyond *n_slots return at once.
  const uint32_t *n_slots;
  uint32_t *slot_cw;
  // Progress word in host-visible (pinned, mapped) memory, or null: the first check-node launch of
  // an iteration publishes (epoch, iteration, codewords still running) there, so that the host can
  // stop enqueuing launches for a group that has finished -- without a stream synchronisation.
  uint64_t *publish;
  uint32_t epoch, tick;
  // Row-record flooding path, or null: per wave slice (64 * VEC codewords) 0 = no codeword of the slice has
  // converged since the group started (nobody needs the posterior of the L-free variables: it is not stored),
  // 1 = the first ones just have (vn_kernel sets it; vn_free_rec_kernel's event mode rebuilds their L-free
  // posteriors from the records), 2 = stored by the check-node kernel every iteration from now on
  uint32_t *slice_state;
  // Continuous batching (DeviceDecoder::decode_stream), or null: the group never drains -- a slot whose codeword
  // has finished is handed a fresh one at the next harvest -- so every slot counts its own iterations:
  // it0[slot] = group iterations completed when the slot's codeword started, max_it = the per-codeword limit
  const uint32_t *it0;
  uint32_t max_it;
};
enum : uint32_t { kNoCodeword = 0xFFFFFFFFu };

// progress word: epoch (24 bits) | iteration (20 bits) | codewords still running (20 bits)
__host__ __device__ inline uint64_t progress_word(uint32_t epoch, uint32_t tick, uint32_t running) {
  return (uint64_t(epoch & 0xFFFFFFu) << 40) | (uint64_t(tick & 0xFFFFFu) << 20) | uint64_t(running & 0xFFFFFu);
}

// top of every check-node / level kernel: true when the whole group has finished
__device__ __forceinline__ bool group_finished(const State &st) {
  const uint32_t running = *st.n_active;
  if (st.publish != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(st.publish, progress_word(st.epoch, st.tick, min(running, 0xFFFFFu)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  return running == 0;
}

__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ float m_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double m_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float m_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double m_max(double a, double b) { return fmax(a, b); }
// transcendentals: glibc-identical (exact_math.h) in both precisions, so every rule matches the CPU bit for bit
#ifdef LDPC_TRIVIAL_MATH
// measurement builds only (tools/ab_variants.sh): the rules' structure with the f32 functions replaced by
// one multiplication each, to read the kernels' instruction count and time WITHOUT the functions.  Wrong results.
__device__ __forceinline__ float m_tanh(float x) { return 0.25f * x; }
__device__ __forceinline__ float m_log(float x) { return 0.5f * x; }
__device__ __forceinline__ float m_exp(float x) { return 0.5f * x; }
__device__ __forceinline__ float m_log1p(float x) { return 0.5f * x; }
#else
__device__ __forceinline__ float m_tanh(float x) { return em::tanhf(x); }
__device__ __forceinline__ float m_log(float x) { return em::logf(x); }
__device__ __forceinline__ float m_exp(float x) { return em::expf(x); }
__device__ __forceinline__ float m_log1p(float x) { return em::log1pf(x); }
#endif
__device__ __forceinline__ double m_tanh(double x) { return em::tanh(x); }
__device__ __forceinline__ double m_log(double x) { return em::log(x); }
__device__ __forceinline__ double m_exp(double x) { return em::exp(x); }
__device__ __forceinline__ double m_log1p(double x) { return em::log1p(x); }
// ln_1p(exp(-a)), a >= 0: the min* correction term (f32: the fused form)
__device__ __forceinline__ double m_corr(double a) { return m_log1p(m_exp(-a)); }
#ifdef LDPC_TRIVIAL_MATH
__device__ __forceinline__ float m_corr(float a) { return 0.25f * a; }
#else
__device__ __forceinline__ float m_corr(float a) { return em::corrf(a); }
#endif

template <typename T>
struct Limits;
template <>
struct Limits<float> {
  static constexpr float tanh_clamp = 9.0f;   // arithmetic.rs:435
  static constexpr float phi_min_x = 1e-30f;  // arithmetic.rs:298
  __device__ static float inf() { return __builtin_huge_valf(); }
};
template <>
struct Limits<double> {
  static constexpr double tanh_clamp = 18.0;  // arithmetic.rs:433
  static constexpr double phi_min_x = 1e-30;  // arithmetic.rs:297
  __device__ static double inf() { return __builtin_huge_val(); }
};

// ---------------------------------------------------------------------------------------
// Check-node rules on an LDS column: x[i*S], out[i*S], scr[i*S] for slot i of this thread.
// ---------------------------------------------------------------------------------------

// arithmetic.rs:180-186
template <typename T>
__device__ __forceinline__ T phi_fn(T x) {
  x = m_max(x, Limits<T>::phi_min_x);
  return -(m_log(m_tanh(T(0.5) * x)));
}
#ifndef LDPC_TRIVIAL_MATH
// f32: the fused form (exact_math.h)
template <>
__device__ __forceinline__ float phi_fn<float>(float x) {
  return em::phif(x);
}
#endif

// Rust std atanh: 0.5 * ln_1p(2x / (1 - x))
// tanh of an argument the Tanh rule has clamped to +-tanh_clamp: f32 takes the branch-free form
__device__ __forceinline__ double m_tanh_clamped(double x) { return m_tanh(x); }
#if defined(LDPC_TRIVIAL_MATH) || defined(LDPC_GENERIC_TANH)
__device__ __forceinline__ float m_tanh_clamped(float x) { return m_tanh(x); }
#else
__device__ __forceinline__ float m_tanh_clamped(float x) { return em::tanhf_c9(x); }
#endif
#ifdef LDPC_TRIVIAL_MATH
__device__ __forceinline__ float atanh_rs(float x) { return 0.5f * x; }
#else
__device__ __forceinline__ float atanh_rs(float x) { return em::atanh_rs(x); }
#endif
__device__ __forceinline__ double atanh_rs(double x) { return 0.5 * m_log1p((2.0 * x) / (1.0 - x)); }
// 2 atanh(p) as the Tanh rule forms it (arithmetic.rs:376).  (The straight-line atanh of the slice kernel --
// exact_math.h, atanh_rs_main, rare arguments redone per wavefront -- was tried here too, where the function is
// evaluated in a loop and exists once: its extra selects cost more than the rarely taken branches of atanh_rs save.
// 5G NR BG1 Zc=384 HLTanhf32 34.5 k against 34.8 k codewords/s, DVB-S2 1/2 Tanhf32 0.440 against 0.452 of the roofline,
// alternating runs on one box, round 4.)
__device__ __forceinline__ double two_atanh(double p) { return 2.0 * atanh_rs(p); }
__device__ __forceinline__ float two_atanh(float p) { return 2.0f * atanh_rs(p); }

// ---- "@fast" (opt-in): the same formulas on the GPU's native v_exp_f32 / v_log_f32 / v_rcp_f32 (about 1 ulp each)
// instead of the glibc-identical functions.  Near the origin, where e^x - 1 and 1 +- p cancel, the odd series is used.
__device__ __forceinline__ float fast_tanh(float h) {  // |h| <= 9
  const float e = __builtin_amdgcn_exp2f(h * 2.8853900817779268f);  // e^(2h)
  const float big = (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
  const float h2 = h * h;
  const float small = h * (1.0f + h2 * (-0.33333333f + h2 * 0.13333333f));
  float t = m_abs(h) < 0.125f ? small : big;
  // never +-1 exactly (tanhf(9) is below 1 in f32 too): the row product stays inside atanh's domain
  return m_max(m_min(t, 0x1.fffffep-1f), -0x1.fffffep-1f);
}
__device__ __forceinline__ float fast_2atanh(float p) {  // |p| < 1: ln((1 + p) / (1 - p))
  const float big = 0.6931471805599453f * (__builtin_amdgcn_logf(1.0f + p) - __builtin_amdgcn_logf(1.0f - p));
  const float p2 = p * p;
  const float small = 2.0f * p * (1.0f + p2 * (0.33333333f + p2 * 0.2f));
  return m_abs(p) < 0.1f ? small : big;
}
__device__ __forceinline__ float fast_phi(float x) {  // -ln(tanh(max(x, 1e-30) / 2)), arithmetic.rs:180-186
  x = m_max(x, 1e-30f);
  return -0.6931471805599453f * __builtin_amdgcn_logf(fast_tanh(m_min(0.5f * x, 9.0f)));
}

// The Tanh rule's exclusion products for a row of exactly D edges: the D values come out of the LDS column in one
// burst and the products are straight-line register arithmetic, in the rule's order (prefix times the tail, slot by slot)
template <typename T, int D>
__device__ __forceinline__ void tanh_products(T *A, uint32_t S) {
  T t[D];
#pragma unroll
  for (int i = 0; i < D; i++) t[i] = A[i * S];
  T prefix = T(1.0);
#pragma unroll
  for (int i = 0; i < D; i++) {
    T product = prefix;
#pragma unroll
    for (int j = i + 1; j < D; j++) product *= t[j];
    prefix *= t[i];
    A[i * S] = product;
  }
}
template <typename T>
__device__ __forceinline__ bool tanh_products_by_degree(T *A, uint32_t d, uint32_t S) {
  switch (d) {
    case 2: tanh_products<T, 2>(A, S); return true;
    case 3: tanh_products<T, 3>(A, S); return true;
    case 4: tanh_products<T, 4>(A, S); return true;
    case 5: tanh_products<T, 5>(A, S); return true;
    case 6: tanh_products<T, 6>(A, S); return true;
    case 7: tanh_products<T, 7>(A, S); return true;
    case 8: tanh_products<T, 8>(A, S); return true;
    case 9: tanh_products<T, 9>(A, S); return true;
    case 10: tanh_products<T, 10>(A, S); return true;
    case 19: tanh_products<T, 19>(A, S); return true;
    default: return false;
  }
}

// Rules work on two LDS columns of the calling thread, A[i*S] and B[i*S]: on entry A holds the
// d inputs x_i in slot order; on return the d outputs are in the column the function returns
// (B, with x intact in A -- except Tanh, which works in A alone and leaves its outputs there).
template <int RULE, typename T>
__device__ __forceinline__ T *rule_check_node(T *A, T *B, uint32_t d, uint32_t S) {
  if constexpr (RULE == kRulePhiFast) {
    // arithmetic.rs:214-246 with fast_phi
    uint32_t sign = 0;
    float sum = 0.0f;
    for (uint32_t i = 0; i < d; i++) {
      const float xi = A[i * S];
      const float p = fast_phi(m_abs(xi));
      B[i * S] = p;
      sum += p;
      if (xi < 0.0f) sign ^= 1u;
    }
    for (uint32_t i = 0; i < d; i++) {
      const float y = fast_phi(sum - B[i * S]);
      const uint32_t s = (A[i * S] < 0.0f) ? (sign ^ 1u) : sign;
      B[i * S] = (s == 0) ? y : -y;
    }
    return B;
  } else if constexpr (RULE == kRuleTanhFast) {
    // arithmetic.rs:347-379 with fast_tanh / fast_2atanh (one column, as the exact Tanh rule)
    for (uint32_t i = 0; i < d; i++) {
      float h = 0.5f * A[i * S];
      h = m_max(m_min(h, 9.0f), -9.0f);
      A[i * S] = fast_tanh(h);
    }
    float prefix = 1.0f;
    for (uint32_t i = 0; i < d; i++) {
      float product = prefix;
      for (uint32_t j = i + 1; j < d; j++) product *= A[j * S];
      prefix *= A[i * S];
      A[i * S] = fast_2atanh(product);
    }
    return A;
  } else if constexpr (RULE == kRulePhi) {
    // arithmetic.rs:214-246
    uint32_t sign = 0;
    T sum = T(0.0);
    for (uint32_t i = 0; i < d; i++) {
      const T xi = A[i * S];
      const T p = phi_fn(m_abs(xi));
      B[i * S] = p;
      sum += p;
      if (xi < T(0.0)) sign ^= 1u;
    }
    for (uint32_t i = 0; i < d; i++) {
      const T y = phi_fn(sum - B[i * S]);
      const uint32_t s = (A[i * S] < T(0.0)) ? (sign ^ 1u) : sign;
      B[i * S] = (s == 0) ? y : -y;
    }
    return B;
  } else if constexpr (RULE == kRuleTanh) {
    // arithmetic.rs:347-379: t_i = tanh(clamp(x_i/2)); out_i = 2 atanh(prod_{j != i} t_j),
    // product from 1.0 in slot order (the O(d^2) order is kept: it fixes the rounding).
    // Everything happens in column A (B is not touched: the launches of this rule allocate one column, which
    // doubles the workgroups per CU for the levels with long rows): x_i is dead once t_i exists, and t_i once
    // the running prefix has absorbed it.
    const T c = Limits<T>::tanh_clamp;
    for (uint32_t i = 0; i < d; i++) {
      T h = T(0.5) * A[i * S];
      if (h < -c) h = -c;  // f32::clamp: a NaN stays a NaN (the reference's arithmetic, tested)
      if (h > c) h = c;
      A[i * S] = m_tanh_clamped(h);
    }
    // prod_{j != i} in slot order from 1.0: the factors before i are the same running prefix for
    // every i (same operations, same rounding), only the tail differs
    // (the common degrees: the products as straight-line register arithmetic after one burst of LDS reads instead
    // of d^2/2 dependent LDS reads -- BG1 Zc=384 HLTanhf32 +4 %, same operations)
    if (tanh_products_by_degree(A, d, S)) {
      for (uint32_t i = 0; i < d; i++) A[i * S] = two_atanh(A[i * S]);
      return A;
    }
    T prefix = T(1.0);
    for (uint32_t i = 0; i < d; i++) {
      T product = prefix;
      for (uint32_t j = i + 1; j < d; j++) product *= A[j * S];
      prefix *= A[i * S];
      A[i * S] = two_atanh(product);
    }
    return A;
  } else if constexpr (RULE == kRuleMinstarapprox || RULE == kRuleMinsum) {
    // arithmetic.rs:487-521 (Minsum: same fold without the correction and the clamp,
    // SURVEY.md Appendix A.6)
    // out_i folds the other inputs in slot order.  The fold over the inputs before i is the same
    // running prefix for every i (identical operations, identical rounding); only the tail is
    // evaluated per output -- half the work of the literal O(d^2) loop, same bits.
    // Minsum folds from +inf with the NaN-ignoring minimum (same value as starting from the first
    // magnitude for non-NaN inputs; matches the streaming kernels when inf - inf produced NaNs)
    uint32_t psign = 0;
    bool phave = RULE == kRuleMinsum;
    T pacc = RULE == kRuleMinsum ? Limits<T>::inf() : T(0.0);
    for (uint32_t i = 0; i < d; i++) {
      uint32_t sign = psign;
      bool have = phave;
      T acc = pacc;
      for (uint32_t j = i + 1; j < d; j++) {
        T v = A[j * S];
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
      B[i * S] = (sign == 0) ? acc : -acc;
      // extend the prefix by input i
      T v = A[i * S];
      if (v < T(0.0)) psign ^= 1u;
      v = m_abs(v);
      if (!phave) {
        pacc = v;
        phave = true;
      } else if constexpr (RULE == kRuleMinsum) {
        pacc = m_min(v, pacc);
      } else {
        pacc = m_max(m_min(v, pacc) - m_corr(m_abs(v - pacc)), T(0.0));
      }
    }
    return B;
  } else {
    // Aminstar, arithmetic.rs:942-999: argmin = FIRST minimum of |x|
    uint32_t argmin = 0;
    T vmin = m_abs(A[0]);
    for (uint32_t i = 1; i < d; i++) {
      const T a = m_abs(A[i * S]);
      if (a < vmin) {
        vmin = a;
        argmin = i;
      }
    }
    uint32_t sign = 0;
    bool have = false;
    T delta = T(0.0);
    for (uint32_t j = 0; j < d; j++) {
      T v = A[j * S];
      if (v < T(0.0)) sign ^= 1u;
      if (j != argmin) {
        v = m_abs(v);
        if (!have) {
          delta = v;
          have = true;
        } else {
          delta = m_min(v, delta) - m_corr(m_abs(v - delta)) + m_corr(v + delta);
        }
      }
    }
    const T xmin = A[argmin * S];
    const T first = ((sign != 0) != (xmin < T(0.0))) ? -delta : delta;
    delta = m_min(delta, vmin) - m_corr(m_abs(delta - vmin)) + m_corr(delta + vmin);
    for (uint32_t j = 0; j < d; j++) {
      const T v = A[j * S];
      B[j * S] = (j == argmin) ? first : (((sign != 0) != (v < T(0.0))) ? -delta : delta);
    }
    return B;
  }
}

// ---------------------------------------------------------------------------------------
// Flooding, min-sum check nodes: streaming kernel, state in registers.
//   L    [N][tile]   posterior of the previous iteration (channel LLRs when FIRST)
//   msg  [E][tile]   check->variable messages, rewritten in place
// v2c is never stored: x = L[v] - msg[e] is the same subtraction the reference's
// variable node performs (arithmetic.rs:152), evaluated here by the consumer.
// The parity of hard(L) over the row is the syndrome bit of the PREVIOUS iteration's
// posterior (flooding.rs:69-79), accumulated per codeword across this wave's rows.
// The graph indices of the NEXT row are fetched (scalar loads) while the current row's
// vector loads are in flight, so a wave's dependent chain per row is one memory latency.
// ---------------------------------------------------------------------------------------
template <typename T, int VEC, typename MASK, int U, bool FIRST, bool NT>
__global__ __launch_bounds__(256) void cn_minsum_kernel(
    Graph g, Sched sc, State st, const T *__restrict__ L, T *__restrict__ msg,
    uint32_t *__restrict__ unsat_out) {
  if (group_finished(st)) return;  // (publishes the progress word when the launch carries one: a paced host follows it)
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const uint32_t *__restrict__ done = st.done;
  const uint32_t n_rows = g.n_rows, waves_per_chunk = sc.waves_per_chunk;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * VEC;  // codeword index (flag arrays)
  const size_t G = sc.tile;                       // row stride inside a tile
  L += tile_base(b0, g.n_cols, sc) + lane * VEC;
  msg += tile_base(b0, g.n_edges, sc) + lane * VEC;
  {
    bool all_done = true;
#pragma unroll
    for (int k = 0; k < VEC; k++) all_done = all_done && (done[off + k] != 0);
    if (__builtin_amdgcn_ballot_w64(!all_done) == 0) return;
  }
  uint32_t odd_acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; k++) odd_acc[k] = 0;

  // indices of the current row: edge range and the variables of its first U edges
  uint32_t c = node0, e0 = 0, e1 = 0, cols[U];
  if (c < n_rows) {
    e0 = row_ptr[c];
    e1 = row_ptr[c + 1];
  }
#pragma unroll
  for (int u = 0; u < U; u++) cols[u] = edge_col[min(e0 + u, g.n_edges - 1)];

  while (c < n_rows) {
    T min1[VEC], min2[VEC];
    uint32_t arg[VEC], par[VEC];
    MASK sgn[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      min1[k] = Limits<T>::inf();
      min2[k] = Limits<T>::inf();
      arg[k] = 0;
      par[k] = 0;
      sgn[k] = 0;
    }
    // next row's edge range: issued now, consumed after this row's loads are in flight
    const uint32_t cn = c + waves_per_chunk;
    uint32_t ne0 = 0, ne1 = 0;
    if (cn < n_rows) {
      ne0 = row_ptr[cn];
      ne1 = row_ptr[cn + 1];
    }
    uint32_t ncols[U];
    for (uint32_t i0 = e0; i0 < e1; i0 += U) {
      Pack<T, VEC> lv[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t e = min(i0 + u, e1 - 1);
        // slots beyond the degree re-read slot 0 / the last edge (cache hits), masked below
        const uint32_t v = (i0 + u < e1) ? ((i0 == e0) ? cols[u] : edge_col[e]) : cols[0];
        lv[u] = load_pack<T, VEC>(L + size_t(v) * G);
        if (!FIRST) mv[u] = load_msg<T, VEC, NT>(msg + size_t(e) * G);
      }
      if (i0 == e0) {
#pragma unroll
        for (int u = 0; u < U; u++) ncols[u] = edge_col[min(ne0 + u, g.n_edges - 1)];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < e1) {
          const uint32_t slot = i0 + u - e0;
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            const T l = lv[u].v[k];
            const T x = FIRST ? l : (l - mv[u].v[k]);
            const T a = m_abs(x);
            if (x < T(0.0)) sgn[k] |= MASK(1) << slot;
            if (l <= T(0.0)) par[k] ^= 1u;
            if (a < min1[k]) {
              min2[k] = min1[k];
              min1[k] = a;
              arg[k] = slot;
            } else if (a < min2[k]) {
              min2[k] = a;
            }
          }
        }
      }
    }
    if (e0 == e1) {  // empty row: nothing loaded, still fetch the next row's variables
#pragma unroll
      for (int u = 0; u < U; u++) ncols[u] = edge_col[min(ne0 + u, g.n_edges - 1)];
    }
    uint32_t tot[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      tot[k] = (sizeof(MASK) == 8 ? __popcll(sgn[k]) : __popc(uint32_t(sgn[k]))) & 1u;
      odd_acc[k] |= par[k];
    }
    const uint32_t d = e1 - e0;
    for (uint32_t slot = 0; slot < d; slot++) {
      Pack<T, VEC> o;
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const uint32_t neg = uint32_t(sgn[k] >> slot) & 1u;
        const T mag = (arg[k] == slot) ? min2[k] : min1[k];
        o.v[k] = (tot[k] ^ neg) ? -mag : mag;
      }
      store_msg<T, VEC, NT>(msg + size_t(e0 + slot) * G, o);
    }
    c = cn;
    e0 = ne0;
    e1 = ne1;
#pragma unroll
    for (int u = 0; u < U; u++) cols[u] = ncols[u];
  }
  if (!FIRST) {
#pragma unroll
    for (int k = 0; k < VEC; k++)
      if (odd_acc[k]) unsat_out[off + k] = 1u;
  }
}

// ---------------------------------------------------------------------------------------
// Flooding min-sum check nodes with L-free variables (Graph::edge_aux): for an edge whose
// variable has degree <= 2 the kernel reads the channel LLR and the variable's other message
// and forms L = chan + (m_own + m_other) itself -- the two-term slot-ordered sum of
// arithmetic.rs:146 is commutative, so this is bit-identical -- then x = L - m_own.  The
// variable's first slot also stores L into `post` (kept for frozen codewords), so `post` is
// always the previous iteration's posterior, exactly as with the plain kernels.  Saves the
// variable-node kernel 4 row accesses per such variable (half of DVB-S2's variables).
// Because a check now reads a neighbour's message, messages are double-buffered: read from
// msg_in (previous iteration), write to msg.
// ---------------------------------------------------------------------------------------
template <typename T, int VEC, typename MASK, int U, bool FIRST, bool NT, bool NT_IN>
__global__ __launch_bounds__(256) void cn_minsum_lfree_kernel(
    Graph g, Sched sc, State st, const T *__restrict__ chan, T *__restrict__ post,
    const T *__restrict__ msg_in, T *__restrict__ msg, uint32_t *__restrict__ unsat_out) {
  if (group_finished(st)) return;  // (publishes the progress word when the launch carries one: a paced host follows it)
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const TablePtr edge_aux = table_ptr(g.edge_aux);
  const uint32_t *__restrict__ done = st.done;
  const uint32_t n_rows = g.n_rows, waves_per_chunk = sc.waves_per_chunk;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * VEC;
  const size_t G = sc.tile;
  chan += tile_base(b0, g.n_cols, sc) + lane * VEC;
  post += tile_base(b0, g.n_cols, sc) + lane * VEC;
  msg += tile_base(b0, g.n_edges, sc) + lane * VEC;
  msg_in += tile_base(b0, g.n_edges, sc) + lane * VEC;
  bool live[VEC];
  bool any_live = false, all_live = true;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    live[k] = done[off + k] == 0;
    any_live = any_live || live[k];
    all_live = all_live && live[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  uint32_t odd_acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; k++) odd_acc[k] = 0;

  for (uint32_t c = node0; c < n_rows; c += waves_per_chunk) {
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    if (e0 == e1) continue;
    T min1[VEC], min2[VEC];
    uint32_t arg[VEC], par[VEC];
    MASK sgn[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      min1[k] = Limits<T>::inf();
      min2[k] = Limits<T>::inf();
      arg[k] = 0;
      par[k] = 0;
      sgn[k] = 0;
    }
    for (uint32_t i0 = e0; i0 < e1; i0 += U) {
      Pack<T, VEC> lv[U], mv[U], mo[U];
      uint32_t aux[U], var[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        aux[u] = kAuxNone;
        var[u] = 0;
        if (i0 + u < e1) {  // wave-uniform
          const uint32_t e = i0 + u;
          var[u] = edge_col[e];
          aux[u] = edge_aux[e];
          if (aux[u] == kAuxNone) {
            lv[u] = load_pack<T, VEC>(post + size_t(var[u]) * G);
          } else {
            lv[u] = load_pack<T, VEC>(chan + size_t(var[u]) * G);
            if (!FIRST && (aux[u] & kAuxMask) != kAuxSingle)
              mo[u] = load_pack<T, VEC>(msg_in + size_t(aux[u] & kAuxMask) * G);  // re-read by the neighbour: keep cached
          }
          if (!FIRST) mv[u] = load_msg<T, VEC, NT_IN>(msg_in + size_t(e) * G);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < e1) {
          const uint32_t slot = i0 + u - e0;
          const bool lfree = aux[u] != kAuxNone;
          const bool single = (aux[u] & kAuxMask) == kAuxSingle;
          Pack<T, VEC> lnew;
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            T l = lv[u].v[k];
            if (lfree && !FIRST) {
              const T ssum = single ? mv[u].v[k] : (mv[u].v[k] + mo[u].v[k]);
              l = l + ssum;  // chan + (m_a + m_b)
            }
            lnew.v[k] = l;
            const T x = FIRST ? l : (l - mv[u].v[k]);
            const T a = m_abs(x);
            if (x < T(0.0)) sgn[k] |= MASK(1) << slot;
            if (l <= T(0.0)) par[k] ^= 1u;
            if (a < min1[k]) {
              min2[k] = min1[k];
              min1[k] = a;
              arg[k] = slot;
            } else if (a < min2[k]) {
              min2[k] = a;
            }
          }
          if (lfree && !FIRST && (aux[u] & kAuxWriter)) {
            T *dst = post + size_t(var[u]) * G;
            if (all_live) {
              store_pack<T, VEC>(dst, lnew);
            } else {
#pragma unroll
              for (int k = 0; k < VEC; k++)
                if (live[k]) dst[k] = lnew.v[k];
            }
          }
        }
      }
    }
    uint32_t tot[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      tot[k] = (sizeof(MASK) == 8 ? __popcll(sgn[k]) : __popc(uint32_t(sgn[k]))) & 1u;
      odd_acc[k] |= par[k];
    }
    const uint32_t d = e1 - e0;
    for (uint32_t slot = 0; slot < d; slot++) {
      Pack<T, VEC> o;
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const uint32_t neg = uint32_t(sgn[k] >> slot) & 1u;
        const T mag = (arg[k] == slot) ? min2[k] : min1[k];
        o.v[k] = (tot[k] ^ neg) ? -mag : mag;
      }
      store_msg<T, VEC, NT>(msg + size_t(e0 + slot) * G, o);
    }
  }
  if (!FIRST) {
#pragma unroll
    for (int k = 0; k < VEC; k++)
      if (odd_acc[k]) unsat_out[off + k] = 1u;
  }
}

// ---------------------------------------------------------------------------------------
// Flooding min-sum check nodes with ROW RECORDS (default for Minsum f32/f64 when the rows fit the record's
// sign word).  A min-sum check row sends only two magnitudes: every c2v of the row is +-min1, except the one
// on the argmin slot, +-min2 (arithmetic.rs:487-521 without the correction: SURVEY.md Appendix A.6).  So the
// row's d messages ARE the record {min1, min2, flip bits, argmin} -- three words (four when d > 26 in f32):
//   c2v(slot) = (slot == argmin ? min2 : min1) with the sign bit  flip[slot] = total sign parity ^ (x_slot < 0),
// bit for bit the value the per-edge kernels store.  This kernel therefore
//   * reads its own previous messages as ONE record instead of d words (DVB-S2 1/2: 3 instead of 7),
//   * for an edge whose variable is L-free (degree <= 2, see cn_minsum_lfree_kernel) rebuilds the variable's
//     other message from the PEER row's record (Graph::edge_peer = peer row | peer slot).  A wavefront walks
//     runs of `run` consecutive rows: in DVB-S2's staircase the peers are rows c-1 and c+1, whose records the
//     same wavefront loads as its own one step earlier / later (cache hits, not HBM traffic),
//   * writes the new record, and per-edge messages ONLY for the edges of the variables the variable-node
//     kernel still walks (degree >= 3): 5 of 7 words for DVB-S2 1/2.
// Records are double-buffered (a row reads its neighbours' previous records while they write their new ones);
// the per-edge messages no longer are (nobody but vn_kernel reads them).  Per row of DVB-S2 1/2 the launch
// moves 3 + 5 + 1 + 3 + 5 + 1 = 18 words where cn_minsum_lfree_kernel moves 22-24.
//   rec_in / rec_out  [M * RECW][tile]  words of T's size: row c occupies rows c*RECW .. c*RECW + RECW-1
// ---------------------------------------------------------------------------------------
template <typename T>
struct RecWord {
  typedef uint32_t type;
  static constexpr int kArgShift = 26;  // RECW == 3: argmin above the flip bits (rows of at most 26 edges)
};
template <>
struct RecWord<double> {
  typedef uint64_t type;
  static constexpr int kArgShift = 58;
};
// edge_peer[e], an edge whose variable the variable-node kernel walks: kPeerKeep | position of its message in `msg`
// (the variable-major order that kernel reads); an edge of an L-free variable: writer << 30 | peer row << 6 | peer
// slot -- where the variable's OTHER message lives (row field kPeerSingle: there is none, degree 1)
enum : uint32_t { kPeerKeep = 0x80000000u, kPeerPosMask = 0x7FFFFFFFu, kPeerWriter = 0x40000000u, kPeerRowMask = 0xFFFFFFu,
                  kPeerSingle = 0xFFFFFFu };

// gfx950 store-data hazard the compiler does not know (found in round 5; tools/mb/store_hazard_repro.hip reproduces it
// stand-alone, profiles/r05_store_hazard.txt has the run): a MUBUF store of more than 64 bits reads its data registers
// AFTER issue.  With a literal soffset a vector instruction that rewrites one of them needs 2 wait states behind the store
// (LLVM's GCNHazardRecognizer pads those); with the soffset in an SGPR -- the form every [row][tile] access here takes -- it
// still needs ONE, but the ISA manuals exempt that form and the hazard recogniser follows them (createsVALUHazard:
// "this hazard only exists if the instruction is not using a register in the soffset field"), so nothing is inserted:
// `buffer_store_dwordx4 v[0:3], v58, s[56:59], s0 offen` followed directly by `v_and_b32 v2, 63, v53` stored the new v2 in
// lanes 12-15 of every 16 in about one store of 200 -- round 4's "element 2 of lanes 12-15 differs from run to run".
// The pad is an instruction that USES the data registers: they stay live up to it, so whatever rewrites them is issued
// behind it -- at least one wait state behind the store -- wherever the scheduler moves things.  The build checks the
// result in the code object itself (tools/mb/store_hazard_scan.py, `make lint`, tests/test_isa_lint.py).
typedef uint32_t store_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_data_pad(const store_u32x4 &data) {
#ifndef LDPC_NO_STORE_PAD
  asm volatile("s_nop 0" ::"v"(data));
#endif
}

// [row][tile] accesses of a whole Pack through a buffer descriptor: SGPR row offset, one constant VGPR lane offset
template <typename T, int VEC, bool NT>
__device__ __forceinline__ Pack<T, VEC> buf_load(const RowBuf &b, uint32_t lane_off, uint32_t row_off) {
  constexpr int kBytes = sizeof(T) * VEC;
  static_assert(kBytes == 4 || kBytes == 8 || kBytes == 16, "pack size");
  if constexpr (kBytes == 4)
    return __builtin_bit_cast(Pack<T, VEC>, __builtin_amdgcn_raw_buffer_load_b32(b.r, lane_off, row_off, NT ? 2 : 0));
  else if constexpr (kBytes == 8)
    return __builtin_bit_cast(Pack<T, VEC>, __builtin_amdgcn_raw_buffer_load_b64(b.r, lane_off, row_off, NT ? 2 : 0));
  else
    return __builtin_bit_cast(Pack<T, VEC>, __builtin_amdgcn_raw_buffer_load_b128(b.r, lane_off, row_off, NT ? 2 : 0));
}
template <typename T, int VEC, bool NT>
__device__ __forceinline__ void buf_store(const RowBuf &b, uint32_t lane_off, uint32_t row_off, const Pack<T, VEC> &x) {
  constexpr int kBytes = sizeof(T) * VEC;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  if constexpr (kBytes == 4)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x), b.r, lane_off, row_off, NT ? 2 : 0);
  else if constexpr (kBytes == 8)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), b.r, lane_off, row_off, NT ? 2 : 0);
  else {
    const u32x4 data = __builtin_bit_cast(u32x4, x);
    __builtin_amdgcn_raw_buffer_store_b128(data, b.r, lane_off, row_off, NT ? 2 : 0);
    store_data_pad(data);
  }
}

template <typename T, int VEC, int RECW>
struct RowRec {
  typedef typename RecWord<T>::type W;
  Pack<T, VEC> min1, min2;
  Pack<W, VEC> flip, arg;  // RECW == 3: `flip` is the whole third word, `arg` unused
  // row_off: byte offset of the record's first row in the wavefront's slice; row_bytes: bytes between rows
  __device__ __forceinline__ void load(const RowBuf &b, uint32_t lane_off, uint32_t row_off, uint32_t row_bytes) {
    min1 = buf_load<T, VEC, false>(b, lane_off, row_off);
    min2 = buf_load<T, VEC, false>(b, lane_off, row_off + row_bytes);
    flip = __builtin_bit_cast(Pack<W, VEC>, buf_load<T, VEC, false>(b, lane_off, row_off + 2 * row_bytes));
    if constexpr (RECW == 4) arg = __builtin_bit_cast(Pack<W, VEC>, buf_load<T, VEC, false>(b, lane_off, row_off + 3 * row_bytes));
  }
  template <bool NT>
  __device__ __forceinline__ void store(const RowBuf &b, uint32_t lane_off, uint32_t row_off, uint32_t row_bytes) const {
    buf_store<T, VEC, NT>(b, lane_off, row_off, min1);
    buf_store<T, VEC, NT>(b, lane_off, row_off + row_bytes, min2);
    buf_store<T, VEC, NT>(b, lane_off, row_off + 2 * row_bytes, __builtin_bit_cast(Pack<T, VEC>, flip));
    if constexpr (RECW == 4) buf_store<T, VEC, NT>(b, lane_off, row_off + 3 * row_bytes, __builtin_bit_cast(Pack<T, VEC>, arg));
  }
  // the message this row sends on `slot` (wave-uniform) to codeword k of the lane.  The magnitudes are never
  // negative (nor NaN: a NaN input never wins a `<`), so OR-ing the sign bit in is exactly the negation.
  __device__ __forceinline__ T value(uint32_t slot, int k) const {
    const W a = RECW == 4 ? arg.v[k] : (flip.v[k] >> RecWord<T>::kArgShift);
    const T mag = (a == W(slot)) ? min2.v[k] : min1.v[k];
    const W sign = (flip.v[k] >> slot) << (8 * sizeof(W) - 1);
    return __builtin_bit_cast(T, __builtin_bit_cast(W, mag) | sign);
  }
};

#ifdef LDPC_REC_WAVES
#define LDPC_REC_OCC __attribute__((amdgpu_waves_per_eu(LDPC_REC_WAVES, 8)))
#else
#define LDPC_REC_OCC
#endif
// U: edges of a row whose data loads are issued together with the next record's (rows longer than U take
// further rounds); the graph tables must be padded by U entries (the index fetch of a row reads U of them).
// Wavefronts walk runs of `run` consecutive rows, even runs upwards and odd runs downwards: the two records at
// a run boundary are then wanted by both neighbours at the same moment (their first steps, or their last),
// so one of the two fetches is a cache hit.
// STREAM (continuous batching): a lane whose codeword starts with this launch (State::it0 == the launch's
// iteration - 1) has no previous messages: its own and its peers' read as +0.0 -- `Qv - 0.0`, the reference's initial
// state -- whatever the record arrays hold from the slot's previous codeword.
// LONG: some row has more than U edges (further rounds of U loads; compiled out otherwise: the extra code costs the
// short-row case 2 % in registers and scheduling).
template <typename T, int VEC, int RECW, int U, bool FIRST, bool NT, bool STREAM = false, bool LONG = true>
__global__ __launch_bounds__(256) LDPC_REC_OCC void cn_minsum_rec_kernel(
    Graph g, Sched sc, State st, const T *__restrict__ chan, T *__restrict__ post, const T *__restrict__ rec_in,
    T *__restrict__ rec_out, T *__restrict__ msg, uint32_t *__restrict__ unsat_out, uint32_t run LDPC_DBG_PARAM(dbg)) {
#ifndef LDPC_EXPERIMENTS
  constexpr uint32_t dbg = 0;
#endif
  typedef typename RecWord<T>::type W;
  if (group_finished(st)) return;  // (publishes the progress word when the launch carries one: a paced host follows it)
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const TablePtr edge_peer = table_ptr(g.edge_peer);
  const uint32_t *__restrict__ done = st.done;
  const uint32_t n_rows = g.n_rows, waves_per_chunk = sc.waves_per_chunk, tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * VEC;
  bool live[VEC];
  bool any_live = false, all_live = true;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    live[k] = done[off + k] == 0;
    any_live = any_live || live[k];
    all_live = all_live && live[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  all_live = __builtin_amdgcn_ballot_w64(!all_live) == 0;  // wave-uniform
  bool fresh[VEC];
#pragma unroll
  for (int k = 0; k < VEC; k++) fresh[k] = STREAM && st.it0[off + k] + 1u == st.tick;
  // Posterior of the L-free variables: stored (by the variable's first slot) only in slices where a codeword has
  // converged before -- as long as none has, nothing reads it (State::slice_state; the first convergences of a
  // slice are served by vn_free_rec_kernel's event mode)
  uint32_t write_post = 1;
  if (st.slice_state != nullptr) {
    write_post = st.slice_state[chunk];
    if (write_post == 1 && node0 == 0 && lane == 0) st.slice_state[chunk] = 2;
  }
  if (FIRST || (dbg & 8u)) write_post = 0;
  // the wavefront's slice of every [row][tile] array behind a buffer descriptor: a row access is an SGPR offset
  const uint32_t row_bytes = tile * uint32_t(sizeof(T)), lane_off = lane * uint32_t(VEC * sizeof(T));
  const uint32_t in_tile = in_tile_of(b0, sc) * uint32_t(sizeof(T));
  const RowBuf b_chan = row_buf(chan + tile_base(b0, g.n_cols, sc), uint64_t(g.n_cols) * row_bytes - in_tile);
  const RowBuf b_post = row_buf(post + tile_base(b0, g.n_cols, sc), uint64_t(g.n_cols) * row_bytes - in_tile);
  const RowBuf b_msg = row_buf(msg + tile_base(b0, g.n_edges, sc), uint64_t(g.n_edges) * row_bytes - in_tile);
  const RowBuf b_rin = row_buf(rec_in + tile_base(b0, g.n_rows * RECW, sc), uint64_t(g.n_rows) * RECW * row_bytes - in_tile);
  const RowBuf b_rout = row_buf(rec_out + tile_base(b0, g.n_rows * RECW, sc), uint64_t(g.n_rows) * RECW * row_bytes - in_tile);
  const uint32_t rec_bytes = RECW * row_bytes;
  uint64_t odd_m[VEC];  // lane masks (SGPR pairs): codeword k of the lane has seen an odd row
#pragma unroll
  for (int k = 0; k < VEC; k++) odd_m[k] = 0;

  for (uint32_t r = node0; r * run < n_rows; r += waves_per_chunk) {
    const uint32_t lo = r * run, hi = min(lo + run, n_rows);
    const uint32_t dir = (r & 1u) ? 0xFFFFFFFFu : 1u;  // +1 / -1 (row numbers wrap: an invalid row is >= n_rows)
    uint32_t c = (r & 1u) ? hi - 1 : lo;
    // own = record of the current row, nxt = record of the row the walk reaches next (this row's peer now, `own`
    // one step later); carry = the message the PREVIOUS row of the walk sent to the variable it shares with this
    // one (it had that value in hand as its own message: the previous row's record need not be kept)
    RowRec<T, VEC, RECW> recA, recB;
    T carry[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) carry[k] = T(0.0);
    uint32_t carry_slot = kAuxNone;  // slot of the previous row whose old message `carry` holds
    uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1], ne0 = 0, ne1 = 0;
    if (c + dir < n_rows) {
      ne0 = row_ptr[c + dir];
      ne1 = row_ptr[c + dir + 1];
    }
    uint32_t cols[U], peers[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      cols[u] = edge_col[e0 + u];
      peers[u] = edge_peer[e0 + u];
    }
    if (!FIRST) recA.load(b_rin, lane_off, c * rec_bytes, row_bytes);

    auto row_step = [&](RowRec<T, VEC, RECW> &own, RowRec<T, VEC, RECW> &nxt) {
      const uint32_t d = e1 - e0, cn = c + dir, cp = c - dir;
      if (!FIRST && cn < n_rows) nxt.load(b_rin, lane_off, cn * rec_bytes, row_bytes);
      Pack<T, VEC> lv[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (uint32_t(u) < d)
          lv[u] = buf_load<T, VEC, false>((peers[u] & kPeerKeep) ? b_post : b_chan, lane_off,
                                          ((dbg & 4u) ? uint32_t(u) : cols[u]) * row_bytes);
      // the next row's indices and the range of the row after it: scalar loads that complete while this row's
      // data is in flight
      uint32_t nne0 = 0, nne1 = 0, ncols[U], npeers[U];
      if (cn < n_rows && cn + dir < n_rows) {
        nne0 = row_ptr[cn + dir];
        nne1 = row_ptr[cn + dir + 1];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        ncols[u] = edge_col[ne0 + u];
        npeers[u] = edge_peer[ne0 + u];
      }
      T min1[VEC], min2[VEC];
      uint32_t arg[VEC];
      W sgn[VEC];
      uint64_t par_m[VEC];
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        min1[k] = Limits<T>::inf();
        min2[k] = Limits<T>::inf();
        arg[k] = 0;
        sgn[k] = 0;
        par_m[k] = 0;
      }
      uint32_t next_carry_slot = kAuxNone;
      T next_carry[VEC];
#pragma unroll
      for (int k = 0; k < VEC; k++) next_carry[k] = T(0.0);  // (read below whether or not an edge has set it)
      // one edge: slot, variable, peer word, the loaded soft value (posterior, or channel LLR for an L-free variable)
      auto edge = [&](uint32_t slot, uint32_t var, uint32_t peer, const Pack<T, VEC> &lvu) {
        const bool lfree = !(peer & kPeerKeep);
        const uint32_t prow = (peer >> 6) & kPeerRowMask, pslot = peer & 63u;
        const bool single = prow == kPeerSingle;
        // the variable's other message (wave-uniform choice of where it comes from)
        T m_other[VEC];
        if (lfree && !FIRST && !single) {
          if (prow == cn) {
#pragma unroll
            for (int k = 0; k < VEC; k++) m_other[k] = nxt.value(pslot, k);
          } else if (prow == cp && pslot == carry_slot) {
#pragma unroll
            for (int k = 0; k < VEC; k++) m_other[k] = carry[k];
          } else {
            RowRec<T, VEC, RECW> far;  // not a neighbour inside the run: fetch the peer's record
            far.load(b_rin, lane_off, prow * rec_bytes, row_bytes);
#pragma unroll
            for (int k = 0; k < VEC; k++) m_other[k] = far.value(pslot, k);
          }
          if constexpr (STREAM) {
#pragma unroll
            for (int k = 0; k < VEC; k++) m_other[k] = fresh[k] ? T(0.0) : m_other[k];
          }
        }
        Pack<T, VEC> lnew;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          T l = lvu.v[k];
          T m_own = T(0.0);
          if (!FIRST) {
            m_own = own.value(slot, k);
            if constexpr (STREAM) m_own = fresh[k] ? T(0.0) : m_own;
            if (lfree) l = l + (single ? m_own : (m_own + m_other[k]));  // chan + (m_a + m_b)
          }
          lnew.v[k] = l;
          if (lfree && !FIRST && prow == cn) next_carry[k] = m_own;
          const T x = FIRST ? l : (l - m_own);
          const T a = m_abs(x);
          if (x < T(0.0)) sgn[k] |= W(1) << slot;
          par_m[k] ^= __builtin_amdgcn_ballot_w64(l <= T(0.0));
          if (a < min1[k]) {
            min2[k] = min1[k];
            min1[k] = a;
            arg[k] = slot;
          } else if (a < min2[k]) {
            min2[k] = a;
          }
        }
        if (lfree && !FIRST && prow == cn) next_carry_slot = slot;
        if (lfree && write_post && (peer & kPeerWriter)) {
          if (all_live) {
            buf_store<T, VEC, false>(b_post, lane_off, var * row_bytes, lnew);
          } else {
#pragma unroll
            for (int k = 0; k < VEC; k++)
              if (live[k]) row_store<T, false>(b_post, lane_off + k * uint32_t(sizeof(T)), var * row_bytes, lnew.v[k]);
          }
        }
      };
#pragma unroll
      for (int u = 0; u < U; u++)
        if (uint32_t(u) < d) edge(u, cols[u], peers[u], lv[u]);
      if constexpr (LONG)
      for (uint32_t i0 = U; i0 < d; i0 += U) {  // rows longer than U: further rounds of U loads in flight
        uint32_t cv[U], pv[U];
        Pack<T, VEC> lw[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          cv[u] = edge_col[e0 + i0 + u];  // (the tables are padded: in bounds)
          pv[u] = edge_peer[e0 + i0 + u];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
          if (i0 + u < d) lw[u] = buf_load<T, VEC, false>((pv[u] & kPeerKeep) ? b_post : b_chan, lane_off, cv[u] * row_bytes);
#pragma unroll
        for (int u = 0; u < U; u++)
          if (i0 + u < d) edge(i0 + u, cv[u], pv[u], lw[u]);
      }
      carry_slot = next_carry_slot;
#pragma unroll
      for (int k = 0; k < VEC; k++) carry[k] = next_carry[k];
      if (d != 0) {
        // the new record: flip[slot] = (parity of all signs) ^ (x_slot < 0)
        RowRec<T, VEC, RECW> out;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const uint32_t tot = (sizeof(W) == 8 ? __popcll(sgn[k]) : __popc(uint32_t(sgn[k]))) & 1u;
          odd_m[k] |= par_m[k];
          out.min1.v[k] = min1[k];
          out.min2.v[k] = min2[k];
          const W fl = tot ? ~sgn[k] : sgn[k];
          if constexpr (RECW == 4) {
            out.flip.v[k] = fl;
            out.arg.v[k] = W(arg[k]);
          } else {
            out.flip.v[k] = (fl & ((W(1) << RecWord<T>::kArgShift) - 1)) | (W(arg[k]) << RecWord<T>::kArgShift);
          }
        }
        // (Round 4 kept this store behind an always-true `run != 0`: with it unconditional two variants returned results that
        // differed from run to run.  Round 5 found why -- the gfx950 store-data hazard described at store_data_pad above, a
        // `v_and_b32 v2, ...` issued right behind `buffer_store_dwordx4 v[0:3], ...` -- so the condition is gone: every 128-bit
        // buffer store carries its pad and the build lints the code object.)
#ifdef LDPC_EXPERIMENTS
        if (!(dbg & 2u))
#endif
        out.template store<NT>(b_rout, lane_off, c * rec_bytes, row_bytes);
        // per-edge messages for the variables the variable-node kernel walks, at the position it reads them from
        auto send = [&](uint32_t slot, uint32_t peer) {
          if (!(peer & kPeerKeep) || (dbg & 1u)) return;  // wave-uniform
          Pack<T, VEC> o;
#pragma unroll
          for (int k = 0; k < VEC; k++) o.v[k] = out.value(slot, k);
          buf_store<T, VEC, NT>(b_msg, lane_off, (peer & kPeerPosMask) * row_bytes, o);
        };
#pragma unroll
        for (int u = 0; u < U; u++)
          if (uint32_t(u) < d) send(u, peers[u]);
        if constexpr (LONG)
          for (uint32_t i = U; i < d; i++) send(i, edge_peer[e0 + i]);
      }
      c = cn;
      e0 = ne0;
      e1 = ne1;
      ne0 = nne0;
      ne1 = nne1;
#pragma unroll
      for (int u = 0; u < U; u++) {
        cols[u] = ncols[u];
        peers[u] = npeers[u];
      }
    };
    // two rows per round: the records alternate between recA and recB, no register copies
    for (uint32_t i = lo; i < hi; i += 2) {
      row_step(recA, recB);
      if (i + 1 < hi) row_step(recB, recA);
    }
  }
  if (!FIRST) {
#pragma unroll
    for (int k = 0; k < VEC; k++)
      if ((odd_m[k] >> lane) & 1ull) unsat_out[off + k] = 1u;
  }
}

// Posterior of the L-free variables from the row records: L = chan + (m_a + m_b), the messages read out of the
// records of the variable's one or two rows (free_rs: row << 6 | slot per edge, kAuxNone = no such edge).
//   event_iteration < 0: after the last iteration (no later check-node pass rebuilds it), for the codewords
//                        still running; frozen codewords are skipped;
//   event_iteration >= 0: after the variable-node pass that latched the FIRST converged codewords of a slice
//                        (State::slice_state == 1) at that iteration count: for exactly those codewords, whose
//                        L-free posteriors the check-node kernel had not been storing.
template <typename T, int VEC, int RECW>
__global__ __launch_bounds__(256) void vn_free_rec_kernel(Graph g, Sched sc, State st, const uint32_t *__restrict__ free_rs_,
                                                          const T *__restrict__ chan, const T *__restrict__ rec,
                                                          T *__restrict__ post, int32_t event_iteration) {
  if (event_iteration < 0 && *st.n_active == 0) return;
  const TablePtr free_var = table_ptr(g.list_var), free_rs = table_ptr(free_rs_);
  const uint32_t lane = threadIdx.x & 63u, tile = sc.tile;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, i0;
  wave_slot(sc, wave, &chunk, &i0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  if (event_iteration >= 0 && st.slice_state[chunk] != 1) return;
  const size_t off = size_t(b0) + lane * VEC;
  const size_t G = tile;
  chan += tile_base(b0, g.n_cols, sc) + lane * VEC;
  post += tile_base(b0, g.n_cols, sc) + lane * VEC;
  const uint32_t row_bytes = tile * uint32_t(sizeof(T)), lane_off = lane * uint32_t(VEC * sizeof(T));
  const RowBuf b_rec = row_buf(rec + tile_base(b0, g.n_rows * RECW, sc),
                               uint64_t(g.n_rows) * RECW * row_bytes - in_tile_of(b0, sc) * uint32_t(sizeof(T)));
  bool live[VEC];  // the codewords this pass writes
  bool any_live = false;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    live[k] = event_iteration < 0 ? st.done[off + k] == 0 : (st.done[off + k] != 0 && st.iters[off + k] == event_iteration);
    any_live = any_live || live[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  for (uint32_t i = i0; i < g.n_list; i += sc.waves_per_chunk) {
    const uint32_t v = free_var[i], a = free_rs[2 * i], b = free_rs[2 * i + 1];
    const Pack<T, VEC> ch = load_pack<T, VEC>(chan + size_t(v) * G);
    RowRec<T, VEC, RECW> ra, rb;
    if (a != kAuxNone) ra.load(b_rec, lane_off, (a >> 6) * RECW * row_bytes, row_bytes);
    if (b != kAuxNone) rb.load(b_rec, lane_off, (b >> 6) * RECW * row_bytes, row_bytes);
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      T sum = -T(0.0);  // arithmetic.rs:146: the slot-ordered sum, from Rust's float Sum identity
      if (a != kAuxNone) sum = sum + ra.value(a & 63u, k);
      if (b != kAuxNone) sum = sum + rb.value(b & 63u, k);
      if (live[k]) post[size_t(v) * G + k] = ch.v[k] + sum;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Flooding, any rule: the check row's d inputs are staged in two LDS columns per thread
// ([slot][thread], conflict-free); global loads and stores are issued U at a time.
// dynamic LDS: 2 * dmax * blockDim.x * sizeof(T)
// ---------------------------------------------------------------------------------------
// SCRATCH (round 5): rows too long for the CU's LDS (2 * dmax * 64 * sizeof(T) > 160 KB: more than 320 edges in f32, 160
// in f64 -- the reference takes any alist, /root/reference/src/sparse.rs:352-389) keep the two columns in a per-wavefront
// region of `scratch` in HBM, [2 * dmax][64] -- the same code, the same order of operations, global instead of LDS
// accesses.  Slow by design (nothing real has such rows); the launch is sized to a few thousand waves.
template <int RULE, typename T, bool FIRST, bool SCRATCH = false>
__global__ void cn_staged_kernel(Graph g, Sched sc, State st, const T *__restrict__ L,
                                 T *__restrict__ msg, uint32_t *__restrict__ unsat_out, uint32_t dmax,
                                 T *__restrict__ scratch = nullptr) {
  constexpr int U = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const uint32_t n_rows = g.n_rows, waves_per_chunk = sc.waves_per_chunk, tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t S = SCRATCH ? 64u : blockDim.x;
  T *A = SCRATCH ? scratch + size_t(wave) * 2u * dmax * 64u + lane : reinterpret_cast<T *>(smem) + threadIdx.x;
  T *B = A + size_t(dmax) * S;
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 64;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane;
  const size_t G = tile;
  L += tile_base(b0, g.n_cols, sc) + lane;
  msg += tile_base(b0, g.n_edges, sc) + lane;
  if (__builtin_amdgcn_ballot_w64(st.done[off] == 0) == 0) return;
  uint32_t odd_acc = 0;
  for (uint32_t c = node0; c < n_rows; c += waves_per_chunk) {
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    const uint32_t d = e1 - e0;
    if (d == 0) continue;
    uint32_t par = 0;
    for (uint32_t i0 = 0; i0 < d; i0 += U) {
      T lv[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < d) {
          const uint32_t v = edge_col[e0 + i0 + u];
          lv[u] = L[size_t(v) * G];
          if (!FIRST) mv[u] = load_msg<T, 1, true>(msg + size_t(e0 + i0 + u) * G).v[0];  // streamed once
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < d) {
          A[(i0 + u) * S] = FIRST ? lv[u] : (lv[u] - mv[u]);
          if (lv[u] <= T(0.0)) par ^= 1u;
        }
      }
    }
    odd_acc |= par;
    const T *out = rule_check_node<RULE, T>(A, B, d, S);
    for (uint32_t i0 = 0; i0 < d; i0 += U) {
#pragma unroll
      for (int u = 0; u < U; u++)
        if (i0 + u < d) {
          Pack<T, 1> ov;
          ov.v[0] = out[(i0 + u) * S];
          store_msg<T, 1, true>(msg + size_t(e0 + i0 + u) * G, ov);
        }
    }
  }
  if (!FIRST && odd_acc) unsat_out[off] = 1u;
}

// ---------------------------------------------------------------------------------------
// Flooding, variable nodes (all float rules share arithmetic.rs:140-156):
//   S = sum of the incoming check messages in cols[v] order, folded from -0.0 (Rust's
//   float Sum identity), L = channel + S.  Only L is written; the consumer recomputes
//   L - m.  Also latches codewords whose previous posterior had a zero syndrome
//   (flooding.rs:69-79): they stop being rewritten from this pass on.
// Index fetches of the next variable overlap the current variable's loads (as in the
// check-node kernel).
// ---------------------------------------------------------------------------------------
template <typename T, int VEC, int U, bool NT, bool LIST>
__global__ __launch_bounds__(256) void vn_kernel(
    Graph g, Sched sc, State st, const T *__restrict__ chan, const T *__restrict__ msg,
    T *__restrict__ post, const uint32_t *__restrict__ unsat_in, uint32_t *__restrict__ unsat_clear,
    int32_t latch_iteration) {
  uint32_t *__restrict__ n_active = st.n_active;
  if (*n_active == 0) return;
  const TablePtr col_ptr = table_ptr(LIST ? g.list_ptr : g.col_ptr);
  const TablePtr col_edge = table_ptr(LIST ? g.list_edge : g.col_edge);
  uint32_t *__restrict__ done = st.done;
  int32_t *__restrict__ iters = st.iters;
  const uint32_t n_cols = LIST ? g.n_list : g.n_cols;  // items to process
  const uint32_t waves_per_chunk = sc.waves_per_chunk, tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, v_first;
  wave_slot(sc, wave, &chunk, &v_first);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * VEC;
  const size_t G = tile;
  chan += tile_base(b0, g.n_cols, sc) + lane * VEC;
  post += tile_base(b0, g.n_cols, sc) + lane * VEC;
  msg += tile_base(b0, g.n_edges, sc) + lane * VEC;
  bool skip[VEC];
  bool any_live = false, any_new = false;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    const bool was_done = done[off + k] != 0;
    const bool converged = !was_done && unsat_in != nullptr && unsat_in[off + k] == 0;
    // continuous batching: the codeword's own iteration count; one that has used all of its iterations without
    // converging fails here and keeps its last posterior (flooding.rs:82-85)
    int32_t own_iterations = latch_iteration;
    bool expired = false;
    if (st.it0 != nullptr) {
      own_iterations = latch_iteration - static_cast<int32_t>(st.it0[off + k]);
      expired = !was_done && !converged && own_iterations >= static_cast<int32_t>(st.max_it);
    }
    skip[k] = was_done || converged || expired;
    any_live = any_live || !skip[k];
    if (v_first == 0) {
      // exactly one wave per slice does the per-codeword bookkeeping
      if (converged || expired) {
        done[off + k] = 1u;
        iters[off + k] = converged ? own_iterations : -1;
        atomicSub(n_active, 1u);
        any_new = true;
      }
      unsat_clear[off + k] = 0u;
    }
  }
  if (v_first == 0 && st.slice_state != nullptr && __builtin_amdgcn_ballot_w64(any_new) != 0 && lane == 0 &&
      st.slice_state[chunk] == 0)
    st.slice_state[chunk] = 1;  // the first convergences of this slice: see State::slice_state
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  bool all = true;
#pragma unroll
  for (int k = 0; k < VEC; k++) all = all && !skip[k];

  const uint32_t last_slot = g.n_edges ? g.n_edges - 1 : 0;
  uint32_t v = v_first, s0 = 0, s1 = 0, ed[U], var = v_first;
  if (v < n_cols) {
    s0 = col_ptr[v];
    s1 = col_ptr[v + 1];
    if (LIST) var = table_ptr(g.list_var)[v];
  }
#pragma unroll
  for (int u = 0; u < U; u++) ed[u] = col_edge[min(s0 + u, last_slot)];

  while (v < n_cols) {
    T sum[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) sum[k] = -T(0.0);
    // (a nontemporal load here -- nobody else reads these channel rows in the list variant -- takes 7 us off this kernel
    // and puts 14 us on the check-node kernel that follows: profiles/r04_vn_kernel.txt)
    const Pack<T, VEC> ch = load_pack<T, VEC>(chan + size_t(var) * G);
    const uint32_t vn = v + waves_per_chunk;
    uint32_t ns0 = 0, ns1 = 0, nvar = vn;
    if (vn < n_cols) {
      ns0 = col_ptr[vn];
      ns1 = col_ptr[vn + 1];
      if (LIST) nvar = table_ptr(g.list_var)[vn];
    }
    uint32_t ned[U];
    for (uint32_t j0 = s0; j0 < s1; j0 += U) {
      Pack<T, VEC> mv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (j0 + u < s1) {  // wave-uniform
          const uint32_t e = (j0 == s0) ? ed[u] : col_edge[j0 + u];
          mv[u] = load_msg<T, VEC, NT>(msg + size_t(e) * G);
        }
      }
      if (j0 == s0) {
#pragma unroll
        for (int u = 0; u < U; u++) ned[u] = col_edge[min(ns0 + u, last_slot)];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (j0 + u < s1) {
#pragma unroll
          for (int k = 0; k < VEC; k++) sum[k] = sum[k] + mv[u].v[k];
        }
      }
    }
    if (s0 == s1) {
#pragma unroll
      for (int u = 0; u < U; u++) ned[u] = col_edge[min(ns0 + u, last_slot)];
    }
    Pack<T, VEC> o;
#pragma unroll
    for (int k = 0; k < VEC; k++) o.v[k] = ch.v[k] + sum[k];
    T *dst = post + size_t(var) * G;
    if (all) {
      store_pack<T, VEC>(dst, o);
    } else {
      // (reading the frozen codewords' values back and storing whole packs instead was measured in round 5: no gain at
      // +2 dB, 0.5 % on the fixed-work pass for the extra branch -- profiles/r05_p2_timeline.txt)
#pragma unroll
      for (int k = 0; k < VEC; k++)
        if (!skip[k]) dst[k] = o.v[k];
    }
    v = vn;
    var = nvar;
    s0 = ns0;
    s1 = ns1;
#pragma unroll
    for (int u = 0; u < U; u++) ed[u] = ned[u];
  }
}

// ---------------------------------------------------------------------------------------
// Layered schedule: one dependency level (rows that share no variable, so their serial
// order in horizontal_layered.rs:105-110 is immaterial).  In-place update of Qv and R.
//   Phi / Aminstar:                 R = out; Qv = x + out      (arithmetic.rs:284-291, 1052-1065)
//   Tanh / Minstarapprox / Minsum:  Qv += out - R; R = out     (arithmetic.rs:423-424, 570-573)
// The update pass re-reads Qv and R (L1/L2 hits: the same wave loaded them a moment ago)
// instead of keeping them in LDS, which would halve the occupancy.
// dynamic LDS: 2 * dmax * blockDim.x * sizeof(T)
// ---------------------------------------------------------------------------------------
// (f64: launched with at most 256 threads; telling the compiler so lifts its register cap from 128, where the 24-edge
// register-resident variants spilled up to 65 registers to scratch.  f32 keeps the default bound: its variants fit.)
#ifndef LDPC_HL_BOUNDS
#define LDPC_HL_BOUNDS(T) __launch_bounds__(sizeof(T) == 8 ? 256 : 1024)
// (register-resident f32 rows of at most 10 edges: 8 waves per SIMD asked for -- 64 registers -- where the compiler by
// itself stops at 67-71 and 7 waves: config 3 35.2k -> 35.9k cw/s fixed work, 328k -> 340k at +2 dB, HLPhif32 +2 %.
// Aminstar and Minstarapprox would spill for no gain (Minstarapprox: 0.211 -> 0.195 of the roofline) and keep the
// compiler's choice, as do the 12-edge variants (up to 17 registers spilled at 64; no BASELINE graph has such levels).
// A 20-edge bucket at 5-6 waves measured equal to the 24-edge one.
// Experiment switch: -DLDPC_HL_REG_WAVES=1 restores the compiler's choice everywhere.)
#ifndef LDPC_HL_REG_WAVES
#define LDPC_HL_REG_WAVES 8
#endif
#define LDPC_HL_REG_BOUNDS(RULE, T, DMAX)                               \
  __launch_bounds__(sizeof(T) == 8 ? 256 : (DMAX <= 12 ? 256 : 1024),   \
                    (sizeof(T) == 4 && DMAX <= 10 && RULE != kRuleAminstar && RULE != kRuleMinstarapprox) ? LDPC_HL_REG_WAVES : 1)
#endif
// (SCRATCH: as in cn_staged_kernel -- rows beyond the LDS take per-wavefront columns in HBM)
template <int RULE, typename T, bool FIRST, bool SCRATCH = false>
__global__ LDPC_HL_BOUNDS(T) void hl_level_kernel(Graph g, Sched sc, State st, const uint32_t *__restrict__ level_rows,
                                uint32_t n_level_rows, T *__restrict__ Q, T *__restrict__ R, uint32_t dmax,
                                T *__restrict__ scratch = nullptr) {
  constexpr int U = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const uint32_t waves_per_chunk = sc.waves_per_chunk, tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t S = SCRATCH ? 64u : blockDim.x;
  T *A = SCRATCH ? scratch + size_t(wave) * 2u * dmax * 64u + lane : reinterpret_cast<T *>(smem) + threadIdx.x;
  T *B = A + size_t(dmax) * S;
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 64;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane;
  const size_t G = tile;
  Q += tile_base(b0, g.n_cols, sc) + lane;
  R += tile_base(b0, g.n_edges, sc) + lane;
  const bool frozen = st.done[off] != 0;
  if (__builtin_amdgcn_ballot_w64(!frozen) == 0) return;
  for (uint32_t idx = node0; idx < n_level_rows; idx += waves_per_chunk) {
    const uint32_t c = table_ptr(level_rows)[idx];
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    const uint32_t d = e1 - e0;
    if (d == 0) continue;
    for (uint32_t i0 = 0; i0 < d; i0 += U) {
      T qv[U], rv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < d) {
          const uint32_t v = edge_col[e0 + i0 + u];
          qv[u] = Q[size_t(v) * G];
          if (!FIRST) rv[u] = R[size_t(e0 + i0 + u) * G];
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (i0 + u < d) A[(i0 + u) * S] = FIRST ? (qv[u] - T(0.0)) : (qv[u] - rv[u]);
    }
    const T *out = rule_check_node<RULE, T>(A, B, d, S);
    if (!frozen) {
      for (uint32_t i0 = 0; i0 < d; i0 += U) {
        T qn[U], on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (i0 + u < d) {
            const uint32_t i = i0 + u;
            const T o = out[i * S];
            on[u] = o;
            if constexpr (RULE == kRulePhi || RULE == kRulePhiFast || RULE == kRuleAminstar) {
              qn[u] = A[i * S] + o;
            } else {
              const uint32_t v = edge_col[e0 + i];
              const T q = Q[size_t(v) * G];
              const T r = FIRST ? T(0.0) : R[size_t(e0 + i) * G];
              qn[u] = q + (o - r);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (i0 + u < d) {
            const uint32_t v = edge_col[e0 + i0 + u];
            R[size_t(e0 + i0 + u) * G] = on[u];
            Q[size_t(v) * G] = qn[u];
          }
        }
      }
    }
  }
}

// hl_level_kernel for levels whose rows have at most DMAX edges: the row's Qv and R values are
// loaded into registers in one burst (all loads of the row in flight together, R nontemporal) and
// kept for the update, so there is no second pass over global memory; only the rule's inputs and
// outputs go through the LDS columns (the rules index them dynamically).  With trivial arithmetic
// the two-pass form takes 275 us per BG1 level where the streaming min-sum kernel takes 80: the
// staged structure -- three short load bursts, then three more for the update, at four waves per
// SIMD -- was the cost, not the transcendental functions.
// The rows come as records (slice_tasks.h, build_level_recs: first edge, degree, the edges' variables, 16 or 32 words
// per row in level order): one scalar load per row where the chain level_rows -> row_ptr -> edge_col took four dependent
// ones, and the record is simply loaded again for the update, so the variables' offsets are not held in scalar
// registers across the rule (at 8 waves per SIMD the compiler otherwise parks them in a vector register's lanes).
// f(i) for a row's slots i in [0, d).  Rows of at most 12 edges: one straight-line block per degree behind a switch
// (the chain of `if (i < d)` blocks made the compiler keep its ten conditions as 64-bit masks in scalar registers, and
// at 8 waves per SIMD it then parks scalar registers in a vector register's lanes).  Longer rows keep the chain: a
// block per degree would be the larger cost there.
template <typename F, int... I>
__device__ __forceinline__ void slots_seq(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int D, typename F>
__device__ __forceinline__ void slots_upto(F &&f) {
  slots_seq(f, std::make_integer_sequence<int, D>{});
}
template <typename F, int... I>
__device__ __forceinline__ void slots_below(uint32_t d, F &&f, std::integer_sequence<int, I...>) {
  ((uint32_t(I) < d ? (f(std::integral_constant<int, I>{}), 0) : 0), ...);
}
template <int DMAX, typename F>
__device__ __forceinline__ void for_slots(uint32_t d, F &&f) {
  if constexpr (DMAX <= 12) {
#define LDPC_DEG_CASE(k) \
  case k:                \
    if constexpr (DMAX >= k) slots_upto<k>(f); \
    break;
    switch (d) {
      LDPC_DEG_CASE(1) LDPC_DEG_CASE(2) LDPC_DEG_CASE(3) LDPC_DEG_CASE(4) LDPC_DEG_CASE(5) LDPC_DEG_CASE(6)
      LDPC_DEG_CASE(7) LDPC_DEG_CASE(8) LDPC_DEG_CASE(9) LDPC_DEG_CASE(10) LDPC_DEG_CASE(11) LDPC_DEG_CASE(12)
      default:
        break;
    }
#undef LDPC_DEG_CASE
  } else {
    slots_below(d, f, std::make_integer_sequence<int, DMAX>{});
  }
}
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef const u32x16 __attribute__((address_space(4))) *RecPtr;
template <int DMAX>
__device__ __forceinline__ uint32_t rec_word(const u32x16 &w0, const u32x16 &w1, int i) {
  return i < 16 ? w0[i & 15] : w1[i & 15];
}
template <int RULE, typename T, int DMAX, bool FIRST>
__global__ LDPC_HL_REG_BOUNDS(RULE, T, DMAX) void hl_level_reg_kernel(Graph g, Sched sc, State st, const uint32_t *__restrict__ level_recs,
                                    uint32_t n_level_rows, T *__restrict__ Q, T *__restrict__ R, uint32_t dmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  constexpr uint32_t kRecVecs = DMAX <= 12 ? 1 : 2;  // 16-word pieces of a record
  const RecPtr recs = (RecPtr)level_recs;
  const uint32_t waves_per_chunk = sc.waves_per_chunk, tile = sc.tile;
  const uint32_t S = blockDim.x;
  T *A = reinterpret_cast<T *>(smem) + threadIdx.x;
  T *B = A + size_t(dmax) * S;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 64;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane;
  const bool frozen = st.done[off] != 0;
  if (__builtin_amdgcn_ballot_w64(!frozen) == 0) return;
  // this wavefront's 64-codeword slice of its layout tile, as two buffers; a row is row_bytes apart
  const uint32_t row_bytes = tile * uint32_t(sizeof(T)), lane_off = lane * uint32_t(sizeof(T));
  const size_t tq = tile_base(b0, g.n_cols, sc), tr = tile_base(b0, g.n_edges, sc);
  const RowBuf Qb = row_buf(Q + tq, uint64_t(g.n_cols) * row_bytes - in_tile_of(b0, sc) * sizeof(T));
  const RowBuf Rb = row_buf(R + tr, uint64_t(g.n_edges) * row_bytes - in_tile_of(b0, sc) * sizeof(T));
  for (uint32_t idx = node0; idx < n_level_rows; idx += waves_per_chunk) {
    u32x16 w0 = recs[idx * kRecVecs], w1 = w0;
    if constexpr (kRecVecs == 2) w1 = recs[idx * kRecVecs + 1];
    const uint32_t d = w0[1];
    if (d == 0) continue;
#ifdef LEVEL_EXP  // timing experiments only (tools/ab_variants.sh): 8 = every row reads the tile's first rows (cache hits)
    if (LEVEL_EXP & 8) {
#pragma unroll
      for (int i = 0; i < DMAX; i++) (i + 2 < 16 ? w0[(i + 2) & 15] : w1[(i + 2) & 15]) = uint32_t(i);
      w0[0] = 0;
    }
#endif
    const uint32_t roff = w0[0] * row_bytes;
    T q[DMAX], r[DMAX];
    for_slots<DMAX>(d, [&](auto slot) {
      constexpr int i = decltype(slot)::value;
      q[i] = row_load<T, false>(Qb, lane_off, rec_word<DMAX>(w0, w1, i + 2) * row_bytes);
      if (!FIRST) r[i] = row_load<T, true>(Rb, lane_off, roff + uint32_t(i) * row_bytes);
    });
    for_slots<DMAX>(d, [&](auto slot) {
      constexpr int i = decltype(slot)::value;
      A[i * S] = FIRST ? (q[i] - T(0.0)) : (q[i] - r[i]);
    });
    const T *out = rule_check_node<RULE, T>(A, B, d, S);
    if (!frozen) {
      // the record again (a scalar-cache hit), through a copy of the index the compiler cannot see through
      uint32_t idx2 = idx;
      asm volatile("" : "+s"(idx2));
      u32x16 u0 = recs[idx2 * kRecVecs], u1 = u0;
      if constexpr (kRecVecs == 2) u1 = recs[idx2 * kRecVecs + 1];
#ifdef LEVEL_EXP  // (8: and the stores go out of the buffers' range)
      const uint32_t sbase = (LEVEL_EXP & 8) ? 0x80000000u : 0u;
#else
      constexpr uint32_t sbase = 0;
#endif
      for_slots<DMAX>(d, [&](auto slot) {
        constexpr int i = decltype(slot)::value;
        const T o = out[i * S];
        T qn;
        if constexpr (RULE == kRulePhi || RULE == kRulePhiFast || RULE == kRuleAminstar)
          qn = A[i * S] + o;
        else
          qn = q[i] + (o - (FIRST ? T(0.0) : r[i]));
        row_store<T, true>(Rb, lane_off, sbase + roff + uint32_t(i) * row_bytes, o);
        row_store<T, false>(Qb, lane_off, sbase + rec_word<DMAX>(u0, u1, i + 2) * row_bytes, qn);
      });
    }
  }
}

// Flooding check nodes (the Tanh rule), rows of at most DMAX edges in registers: cn_staged_kernel with hl_level_reg_kernel's row
// handling -- one record per row (slice_tasks.h, build_level_recs over all rows in order: first edge, degree, variables),
// the row's posterior and message values loaded in one burst through buffer descriptors, a straight-line block per degree.
// Same arithmetic per row as cn_staged_kernel (flooding.rs:95-127): x_i = L - c2v_old (the channel value in the first
// iteration), parity of the hard decisions, rule, new messages.
#ifndef LDPC_CN_REG_WAVES
#define LDPC_CN_REG_WAVES 8
#endif
#define LDPC_CN_REG_BOUNDS(T, DMAX) __launch_bounds__(256, (sizeof(T) == 4 && DMAX <= 10) ? LDPC_CN_REG_WAVES : 1)
template <int RULE, typename T, int DMAX, bool FIRST>
__global__ LDPC_CN_REG_BOUNDS(T, DMAX) void cn_reg_kernel(Graph g, Sched sc, State st, const uint32_t *__restrict__ row_recs,
                              const T *__restrict__ L, T *__restrict__ msg, uint32_t *__restrict__ unsat_out, uint32_t dmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  constexpr uint32_t kRecVecs = DMAX <= 12 ? 1 : 2;
  const RecPtr recs = (RecPtr)row_recs;
  const uint32_t n_rows = g.n_rows, waves_per_chunk = sc.waves_per_chunk, tile = sc.tile;
  const uint32_t S = blockDim.x;
  T *A = reinterpret_cast<T *>(smem) + threadIdx.x;
  T *B = A + size_t(dmax) * S;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * 64;
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane;
  if (__builtin_amdgcn_ballot_w64(st.done[off] == 0) == 0) return;
  const uint32_t row_bytes = tile * uint32_t(sizeof(T)), lane_off = lane * uint32_t(sizeof(T));
  const size_t tl = tile_base(b0, g.n_cols, sc), tm = tile_base(b0, g.n_edges, sc);
  const RowBuf Lb = row_buf(L + tl, uint64_t(g.n_cols) * row_bytes - in_tile_of(b0, sc) * sizeof(T));
  const RowBuf Mb = row_buf(msg + tm, uint64_t(g.n_edges) * row_bytes - in_tile_of(b0, sc) * sizeof(T));
  uint32_t odd_acc = 0;
  for (uint32_t c = node0; c < n_rows; c += waves_per_chunk) {
    u32x16 w0 = recs[c * kRecVecs], w1 = w0;
    if constexpr (kRecVecs == 2) w1 = recs[c * kRecVecs + 1];
    const uint32_t d = w0[1];
    if (d == 0) continue;
    const uint32_t moff = w0[0] * row_bytes;
    T lv[DMAX], mv[DMAX];
    for_slots<DMAX>(d, [&](auto slot) {
      constexpr int i = decltype(slot)::value;
      lv[i] = row_load<T, false>(Lb, lane_off, rec_word<DMAX>(w0, w1, i + 2) * row_bytes);
      if (!FIRST) mv[i] = row_load<T, true>(Mb, lane_off, moff + uint32_t(i) * row_bytes);  // streamed once
    });
    uint32_t par = 0;
    for_slots<DMAX>(d, [&](auto slot) {
      constexpr int i = decltype(slot)::value;
      A[i * S] = FIRST ? lv[i] : (lv[i] - mv[i]);
      if (lv[i] <= T(0.0)) par ^= 1u;
    });
    odd_acc |= par;
    const T *out = rule_check_node<RULE, T>(A, B, d, S);
    for_slots<DMAX>(d, [&](auto slot) {
      constexpr int i = decltype(slot)::value;
      row_store<T, true>(Mb, lane_off, moff + uint32_t(i) * row_bytes, out[i * S]);
    });
  }
  if (!FIRST && odd_acc) unsat_out[off] = 1u;
}

#ifdef LDPC_EXPERIMENTS  // opt-in form measured level with / behind the per-level launches (round 4): kept out of the product
// ---------------------------------------------------------------------------------------
// Layered schedule, slice-persistent form: ONE launch per iteration instead of one per dependency level.
// Codewords are independent and the levels only order work inside a codeword (horizontal_layered.rs:105-110), so a
// WORKGROUP owns a slice of SLICE codewords and walks all levels by itself, its waves sharing each level's rows, with a
// workgroup barrier between levels -- no kernel boundary (drain + dispatch, ~17 us each, 32 per iteration on 5G NR
// BG1) and no chip-wide tail per level.  A wavefront takes 64 / SLICE rows of equal degree at a time ("task": lanes
// [k * SLICE, (k + 1) * SLICE) work on the task's k-th row for the slice's codewords), so 8192 codewords in slices of
// 32 are 256 workgroups: one per CU, 16 waves each.  Row accesses stay whole 128-byte lines.
//
// One workgroup per CU means 4 waves per SIMD and nothing else to hide memory latency behind, so the kernel is
// software-pipelined: while a wave computes task t, the Qv and R values of its NEXT task are already in flight into a
// second register set, and the variable indices of the task after that into a third.  A task record has a fixed size
// (row offsets and degree: scalar loads; kSliceD indices per row: one vector load per lane and four indices).
// The register sets hold kSliceD edges per lane: a longer row is SPLIT between the two half-waves (Tanh rule, slices
// of 32: lanes 0-31 take the first half of the row's edges, lanes 32-63 the rest, for the same 32 codewords; the
// row's tanh values meet in one LDS column and every lane forms the exclusion products it needs from all of them).
// Graphs with rows that fit neither way keep the per-level launches (the host decides).  Tasks are handed out through
// one LDS ticket counter per iteration (a wave holds two tickets ahead; tickets past a level's end belong to later
// levels), so the waves of a workgroup stay balanced whatever the rows' degrees.  Qv written by one wave in level l is
// read by another wave of the SAME workgroup (same CU, same L1) after the barrier: workgroup scope is enough, and a
// next-task prefetch never crosses a level boundary.  Arithmetic per row: exactly hl_level_reg_kernel's (the same rule
// functions on the same LDS columns; the Tanh form below multiplies the same factors in the same order).
//   tasks:     [n_tasks + 1][4 + RPT * kSliceW] words (RPT = 64 / SLICE rows): first edge of each row (kNoRow: none;
//              word 1 unused when RPT = 1), degree | flags, 0, then per row kSliceW (>= kSliceD) variable indices
//   task_ptr:  [n_levels + 1] first task of every level
// dynamic LDS: (columns * dmax * sizeof(T) + 2 * kSliceD * 4) * THREADS + 16 bytes (rule columns, parked Qv offsets,
// ticket counter); dmax >= kSliceD
// ---------------------------------------------------------------------------------------
enum : uint32_t { kNoRow = 0xFFFFFFFFu, kTaskSplit = 0x80000000u, kTaskDegMask = 0xFFFFu,
                  kSlicePad = 0x003FFFFFu };  // padding index of a task record: times a row's bytes it is out of every range
constexpr int kSliceD = 10;  // edges per lane and task
constexpr int kSliceW = 12;  // index words per row in a task record (16-byte pieces)

// The Tanh rule (arithmetic.rs:347-379) on registers, for the slice kernel: t[i] = tanh(clamp(x_i / 2)) per edge, then
// the exclusion products -- prod_{j != i} from 1.0 in slot order: the factors before i are the running prefix (the same
// operations, hence the same rounding, for every i), then the tail -- then 2 atanh(.) per edge.
template <int RULE, typename T>
__device__ __forceinline__ T slice_tanh(T q, T r, bool first) {
  const T c = Limits<T>::tanh_clamp;
  T h = T(0.5) * (first ? (q - T(0.0)) : (q - r));
  if constexpr (RULE == kRuleTanhFast) {
    return fast_tanh(m_max(m_min(h, c), -c));
  } else {
    if (h < -c) h = -c;  // f32::clamp: a NaN stays a NaN
    if (h > c) h = c;
    return m_tanh_clamped(h);
  }
}
template <int RULE, typename T>
__device__ __forceinline__ T slice_2atanh(T p) {
  if constexpr (RULE == kRuleTanhFast)
    return fast_2atanh(p);
  else
    return T(2.0) * atanh_rs(p);
}
template <typename T, int D, int N>
__device__ __forceinline__ void tanh_products_reg(T (&t)[N]) {
  T out[D];
  T prefix = T(1.0);
#pragma unroll
  for (int i = 0; i < D; i++) {
    T product = prefix;
#pragma unroll
    for (int j = i + 1; j < D; j++) product *= t[j];
    prefix *= t[i];
    out[i] = product;
  }
#pragma unroll
  for (int i = 0; i < D; i++) t[i] = out[i];
}
// a row of D factors shared by two lanes: all factors from the LDS column; into p[] the products of this lane's
// slots -- [0, ceil(D / 2)) for the first lane, the LAST ceil(D / 2) slots for the second (an odd row's middle slot is
// done by both lanes: the same values twice)
template <typename T, int D, int N, uint32_t S>
__device__ __forceinline__ void tanh_products_shared(const T *A0, bool second, T (&p)[N]) {
  constexpr int kHalf = (D + 1) / 2, kOff = D - kHalf;
  T t[D], pa[N], pb[N];
#pragma unroll
  for (int i = 0; i < D; i++) t[i] = A0[i * S];
  T prefix = T(1.0);
#pragma unroll
  for (int i = 0; i < D; i++) {
    T product = prefix;
#pragma unroll
    for (int j = i + 1; j < D; j++) product *= t[j];
    prefix *= t[i];
    if (i < kHalf && i < N) pa[i] = product;
    if (i >= kOff && i - kOff < N) pb[i - kOff] = product;
  }
#pragma unroll
  for (int k = 0; k < kHalf && k < N; k++) p[k] = second ? pb[k] : pa[k];
}

template <int RULE, typename T, int SLICE, int THREADS, bool FIRST>
__global__ __launch_bounds__(THREADS) void hl_slice_kernel(Graph g, State st, const uint32_t *__restrict__ tasks_,
                                                           const uint32_t *__restrict__ task_ptr_, uint32_t n_levels,
                                                           uint32_t tile, T *__restrict__ Q, T *__restrict__ R,
                                                           uint32_t dmax, uint32_t columns) {
  constexpr uint32_t RPT = 64 / SLICE, TW = 4 + RPT * kSliceW, S = THREADS;
  constexpr bool kTanh = RULE == kRuleTanh || RULE == kRuleTanhFast;
  constexpr uint32_t kOut = 0x80000000u;  // an offset no array of a slice reaches: the access is dropped (range check)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (group_finished(st)) return;
  const uint32_t b0 = blockIdx.x * SLICE;
  if (b0 >= *st.n_slots) return;
  const uint32_t lane = threadIdx.x & 63u, sub = lane / SLICE, cwl = lane % SLICE;
  const bool frozen = st.done[b0 + cwl] != 0;
  // every wave sees the same codewords (the sub-slices repeat them): the exit is workgroup-uniform
  if (__builtin_amdgcn_ballot_w64(!frozen) == 0) return;
  T *A = reinterpret_cast<T *>(smem) + threadIdx.x;
  T *B = A + size_t(dmax) * S;
  // a row shared by the two lanes of a codeword lives in the column of the first one
  T *A0 = reinterpret_cast<T *>(smem) + (threadIdx.x & ~uint32_t(SLICE & 63));
  // the Qv offsets of a task wait in LDS from the issue of its loads to its stores, two tasks' worth (the registers
  // they were computed in take the indices of the task after next meanwhile)
  uint32_t *V = reinterpret_cast<uint32_t *>(smem + size_t(columns) * dmax * S * sizeof(T)) + threadIdx.x;
  uint32_t *ticket = reinterpret_cast<uint32_t *>(smem + size_t(columns) * dmax * S * sizeof(T) + size_t(2 * kSliceD) * S * 4);
  if (threadIdx.x == 0) *ticket = 0;
  __syncthreads();
  const TablePtr tasks = table_ptr(tasks_), task_ptr = table_ptr(task_ptr_);
  const uint32_t n_tasks = task_ptr[n_levels];
  const uint32_t row_bytes = tile * uint32_t(sizeof(T));
  // The memory accesses of a task are branch-free -- always kSliceD loads and kSliceD stores per array, so that the
  // compiler's wait counts are exact (a wave then waits for the loads it issued a task ago, not for the stores it
  // issued a moment ago) -- and what must not happen is pushed out of range instead: a frozen codeword's lane offset,
  // the row offset of a lane without a row, the padding indices of a record (kSlicePad), and the R descriptor of
  // the slots behind the row's last.
#if defined(SLICE_EXP) && (SLICE_EXP & 8)
  const uint32_t lane_off = cwl * uint32_t(sizeof(T)) | kOut;  // timing experiment: every access out of range (no traffic)
#else
  const uint32_t lane_off = cwl * uint32_t(sizeof(T)) | (frozen ? kOut : 0u);
#endif
  const size_t tq = tile_base(b0, g.n_cols, tile), tr = tile_base(b0, g.n_edges, tile);
  const RowBuf Qb = row_buf(Q + tq, uint64_t(g.n_cols) * row_bytes - (b0 % tile) * sizeof(T));
  const RowBuf Rb = row_buf(R + tr, uint64_t(g.n_edges) * row_bytes - (b0 % tile) * sizeof(T));
  const RowBuf Rnone = row_buf(R + tr, 0);
  auto r_buf = [&](bool live) { return live ? Rb : Rnone; };  // wave-uniform: a scalar select of the descriptor
  auto grab = [&]() {
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return uniform(t);
  };
  // what a lane does in a task: its row's (or its part of the row's) first edge as an R offset, how many register
  // slots the wave steps through, and where its slots start in a shared row
  struct Part {
    uint32_t roff, steps, slot0;
  };
  auto part_of = [&](uint32_t e0a, uint32_t e0b, uint32_t info) {
    Part p;
    uint32_t e0 = e0a;
    if constexpr (RPT > 1) e0 = sub ? e0b : e0a;
    p.roff = e0 != kNoRow ? e0 * row_bytes + lane_off : kOut;
    p.steps = info & kTaskDegMask;
    p.slot0 = 0;
    if constexpr (kTanh && RPT == 2) {
      if (info & kTaskSplit) {
        const uint32_t d = p.steps;
        p.steps = (d + 1) / 2;
        p.slot0 = sub ? d - p.steps : 0u;
      }
    }
    return p;
  };
  // per lane: the indices of its row in task t (the record behind the last task is all padding: tickets past the end)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  static_assert(kSliceD == 10 && kSliceW == 12, "fetch_idx reads 4 + 4 + 2 words");
  auto fetch_idx = [&](uint32_t t, uint32_t (&idx)[kSliceD]) {
    const uint32_t *p = tasks_ + size_t(min(t, n_tasks)) * TW + 4 + sub * kSliceW;
    const u32x4 a = *reinterpret_cast<const u32x4 *>(p), b = *reinterpret_cast<const u32x4 *>(p + 4);
    const u32x2 c = *reinterpret_cast<const u32x2 *>(p + 8);
    idx[0] = a.x, idx[1] = a.y, idx[2] = a.z, idx[3] = a.w;
    idx[4] = b.x, idx[5] = b.y, idx[6] = b.z, idx[7] = b.w;
    idx[8] = c.x, idx[9] = c.y;
  };
  // second register set: Qv and R of the wave's next task; idxn: the indices of the task after that (turned into the
  // Qv offsets in place when its loads are issued, parked in V, and overwritten by the following task's indices)
  uint32_t nroff = kOut, idxn[kSliceD];
  T nq[kSliceD], nr[kSliceD];
#pragma unroll
  for (int i = 0; i < kSliceD; i++) {
    nq[i] = T(0.0);
    nr[i] = T(0.0);
  }
  auto issue = [&](uint32_t e0a, uint32_t e0b, uint32_t info, uint32_t (&idx)[kSliceD], uint32_t *park) {
    const Part p = part_of(e0a, e0b, info);
    nroff = p.roff;
#pragma unroll
    for (int i = 0; i < kSliceD; i++) idx[i] = idx[i] * row_bytes + lane_off;
#pragma unroll
    for (int i = 0; i < kSliceD; i++) {
      nq[i] = row_load<T, false>(Qb, idx[i], 0);
      if (!FIRST) nr[i] = row_load<T, true>(r_buf(uint32_t(i) < p.steps), nroff, uint32_t(i) * row_bytes);
    }
#pragma unroll
    for (int i = 0; i < kSliceD; i++) park[i * S] = idx[i];
  };
  auto meta = [&](uint32_t t, uint32_t *e0a, uint32_t *e0b, uint32_t *info) {
    const TablePtr p = tasks + size_t(min(t, n_tasks)) * TW;
    *e0a = p[0];
    *e0b = p[1];
    *info = p[2];
  };
#ifdef SLICE_COUNT
  const uint64_t clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  uint32_t t = grab(), tn = grab();
  uint32_t n_e0a, n_e0b, n_info;  // the record of tn
  meta(tn, &n_e0a, &n_e0b, &n_info);
  fetch_idx(tn, idxn);
  uint32_t par = 0;  // which half of V holds the offsets of the task being computed
  for (uint32_t l = 0; l < n_levels; l++) {
    const uint32_t t1 = task_ptr[l + 1];
    uint32_t c_e0a = 0, c_e0b = 0, c_info = 0;
    if (t < t1) {
      // the wave's first task of this level: nothing of it could be in flight before the barrier
      uint32_t idx[kSliceD];
      meta(t, &c_e0a, &c_e0b, &c_info);
      fetch_idx(t, idx);
      issue(c_e0a, c_e0b, c_info, idx, V + size_t(par * kSliceD) * S);
    }
    while (t < t1) {
      T q[kSliceD], r[kSliceD];
#pragma unroll
      for (int i = 0; i < kSliceD; i++) {
        q[i] = nq[i];
        r[i] = nr[i];
      }
      const uint32_t info = c_info;
      const Part p = part_of(c_e0a, c_e0b, info);
      const uint32_t tn2 = grab();
      // the next task's loads travel while this one computes (never across a level boundary: the rows of the next
      // level read what this level writes -- an all-padding issue keeps the count of memory operations the same)
      uint32_t *park = V + size_t((par ^ 1u) * kSliceD) * S;
      if (tn < t1) {
        issue(n_e0a, n_e0b, n_info, idxn, park);
      } else {
#pragma unroll
        for (int i = 0; i < kSliceD; i++) idxn[i] = kSlicePad;
        issue(kNoRow, kNoRow, 0u, idxn, park);
      }
      c_e0a = n_e0a;
      c_e0b = n_e0b;
      c_info = n_info;
      meta(tn2, &n_e0a, &n_e0b, &n_info);
      fetch_idx(tn2, idxn);
      const uint32_t d = info & kTaskDegMask;
      T o[kSliceD], qn[kSliceD];  // the new messages and posteriors of this lane's slots
      if constexpr (kTanh) {
        // everything in registers; only a shared row's tanh values cross lanes (LDS column A0)
        // (two slots per block where the count allows: a wave issues a dependent chain at half the rate of two
        // interleaved ones, and with four waves per SIMD nothing else fills the gaps)
#pragma unroll
        for (int i = 0; i < kSliceD; i++) o[i] = T(0.0);
#pragma unroll
        for (int i = 0; i < kSliceD; i += 2) {
          if (uint32_t(i + 1) < p.steps) {
            o[i] = slice_tanh<RULE, T>(q[i], r[i], FIRST);
            o[i + 1] = slice_tanh<RULE, T>(q[i + 1], r[i + 1], FIRST);
          } else if (uint32_t(i) < p.steps) {
            o[i] = slice_tanh<RULE, T>(q[i], r[i], FIRST);
          }
        }
#if !(defined(SLICE_EXP) && (SLICE_EXP & 1))
        bool shared = false;
        if constexpr (RPT == 2) shared = (info & kTaskSplit) != 0;
        if (!shared) {
          switch (d) {
            case 2: tanh_products_reg<T, 2>(o); break;
            case 3: tanh_products_reg<T, 3>(o); break;
            case 4: tanh_products_reg<T, 4>(o); break;
            case 5: tanh_products_reg<T, 5>(o); break;
            case 6: tanh_products_reg<T, 6>(o); break;
            case 7: tanh_products_reg<T, 7>(o); break;
            case 8: tanh_products_reg<T, 8>(o); break;
            case 9: tanh_products_reg<T, 9>(o); break;
            case 10: tanh_products_reg<T, 10>(o); break;
            default: o[0] = T(1.0); break;  // one edge: the empty product
          }
        } else {
          T *As = A0 + size_t(p.slot0) * S;
#pragma unroll
          for (int i = 0; i < kSliceD; i++)
            if (uint32_t(i) < p.steps) As[i * S] = o[i];
          if (d == 19) {
            tanh_products_shared<T, 19, kSliceD, S>(A0, sub != 0, o);
          } else {
            T prefix = T(1.0);
            for (uint32_t i = 0; i < d; i++) {
              T product = prefix;
              for (uint32_t j = i + 1; j < d; j++) product *= A0[j * S];
              prefix *= A0[i * S];
#pragma unroll
              for (int k = 0; k < kSliceD; k++)
                if (i == p.slot0 + uint32_t(k)) o[k] = product;
            }
          }
        }
        // 2 atanh(.) per slot: the straight-line form; a slot where some lane holds one of its rare arguments (a few dozen
        // floats inside (-1, 1), and everything outside) is parked in the lane's LDS column and redone with the
        // complete function afterwards -- one copy of that code instead of one per slot
        uint32_t redo = 0;
        auto atanh_slot = [&](int i, T x, T *y) {
          if constexpr (RULE == kRuleTanhFast) {
            *y = fast_2atanh(x);
          } else if constexpr (sizeof(T) == 4) {
            bool rare;
            *y = T(2.0) * em::atanh_rs_main(x, &rare);
            if (__builtin_amdgcn_ballot_w64(rare) != 0) {
              A[i * S] = x;
              redo |= 1u << i;
            }
          } else {
            *y = T(2.0) * atanh_rs(x);
          }
        };
#pragma unroll
        for (int i = 0; i < kSliceD; i += 2) {
          if (uint32_t(i + 1) < p.steps) {
            T y0, y1;
            atanh_slot(i, o[i], &y0);
            atanh_slot(i + 1, o[i + 1], &y1);
            o[i] = y0;
            o[i + 1] = y1;
          } else if (uint32_t(i) < p.steps) {
            T y0;
            atanh_slot(i, o[i], &y0);
            o[i] = y0;
          }
        }
        if (redo != 0) {
          for (uint32_t i = 0; i < p.steps; i++)
            if ((redo >> i) & 1u) A[i * S] = T(2.0) * atanh_rs(A[i * S]);
#pragma unroll
          for (int i = 0; i < kSliceD; i++)
            if ((redo >> i) & 1u) o[i] = A[i * S];
        }
#endif
#pragma unroll
        for (int i = 0; i < kSliceD; i++) qn[i] = q[i] + (o[i] - (FIRST ? T(0.0) : r[i]));
      } else {
#pragma unroll
        for (int i = 0; i < kSliceD; i++) {
          o[i] = T(0.0);
          qn[i] = T(0.0);
          if (uint32_t(i) < p.steps) A[i * S] = FIRST ? (q[i] - T(0.0)) : (q[i] - r[i]);
        }
        const T *out = rule_check_node<RULE, T>(A, B, d, S);
#pragma unroll
        for (int i = 0; i < kSliceD; i++) {
          if (uint32_t(i) < p.steps) {
            o[i] = out[i * S];
            if constexpr (RULE == kRulePhi || RULE == kRulePhiFast || RULE == kRuleAminstar)
              qn[i] = A[i * S] + o[i];
            else
              qn[i] = q[i] + (o[i] - (FIRST ? T(0.0) : r[i]));
          }
        }
      }
#if !(defined(SLICE_EXP) && (SLICE_EXP & 2))
      uint32_t voff[kSliceD];
#pragma unroll
      for (int i = 0; i < kSliceD; i++) voff[i] = V[(par * kSliceD + i) * S];
      par ^= 1u;
#pragma unroll
      for (int i = 0; i < kSliceD; i++) {
        row_store<T, true>(r_buf(uint32_t(i) < p.steps), p.roff, uint32_t(i) * row_bytes, o[i]);
        row_store<T, false>(Qb, voff[i], 0, qn[i]);
      }
#endif
#ifdef SLICE_COUNT
      if (lane == 0) atomicAdd(const_cast<uint32_t *>(st.n_slots) + 2 + (l & 31), 1u);
#endif
      t = tn;
      tn = tn2;
    }
#if !(defined(SLICE_EXP) && (SLICE_EXP & 4))
    __syncthreads();
#endif
  }
#ifdef SLICE_COUNT
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint64_t *w = reinterpret_cast<uint64_t *>(const_cast<uint32_t *>(st.n_slots) + 40);
    w[0] = __builtin_readcyclecounter() - clk0;
    w[1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}
#endif  // LDPC_EXPERIMENTS (slice-persistent layered kernel)

// Layered min-sum (HLMinsumf32/f64, new rule): streaming form of hl_level_kernel, state in
// registers, VEC codewords per lane.  Pass 1 folds min1/min2/first-argmin/sign parity over
// x_i = Qv - R; pass 2 re-reads Qv and R (cache hits), rebuilds x_i, and writes
// R = out, Qv = Qv + (out - R).
template <typename T, int VEC, int U, bool FIRST>
__global__ __launch_bounds__(256) void hl_minsum_kernel(Graph g, Sched sc, State st,
                                                        const uint32_t *__restrict__ level_rows,
                                                        uint32_t n_level_rows, T *__restrict__ Q,
                                                        T *__restrict__ R) {
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const uint32_t waves_per_chunk = sc.waves_per_chunk;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * VEC;
  const size_t G = sc.tile;
  Q += tile_base(b0, g.n_cols, sc) + lane * VEC;
  R += tile_base(b0, g.n_edges, sc) + lane * VEC;
  bool frozen[VEC];
  bool any_live = false;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    frozen[k] = st.done[off + k] != 0;
    any_live = any_live || !frozen[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  bool all_live = true;
#pragma unroll
  for (int k = 0; k < VEC; k++) all_live = all_live && !frozen[k];

  for (uint32_t idx = node0; idx < n_level_rows; idx += waves_per_chunk) {
    const uint32_t c = table_ptr(level_rows)[idx];
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    if (e0 == e1) continue;
    T min1[VEC], min2[VEC];
    uint32_t arg[VEC], tot[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      min1[k] = Limits<T>::inf();
      min2[k] = Limits<T>::inf();
      arg[k] = 0;
      tot[k] = 0;
    }
    for (uint32_t i0 = e0; i0 < e1; i0 += U) {
      Pack<T, VEC> qv[U], rv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < e1) {
          const uint32_t v = edge_col[i0 + u];
          qv[u] = load_pack<T, VEC>(Q + size_t(v) * G);
          if (!FIRST) rv[u] = load_pack<T, VEC>(R + size_t(i0 + u) * G);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < e1) {
          const uint32_t slot = i0 + u - e0;
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            const T x = FIRST ? (qv[u].v[k] - T(0.0)) : (qv[u].v[k] - rv[u].v[k]);
            const T a = m_abs(x);
            if (x < T(0.0)) tot[k] ^= 1u;
            if (a < min1[k]) {
              min2[k] = min1[k];
              min1[k] = a;
              arg[k] = slot;
            } else if (a < min2[k]) {
              min2[k] = a;
            }
          }
        }
      }
    }
    for (uint32_t i0 = e0; i0 < e1; i0 += U) {
      Pack<T, VEC> qv[U], rv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < e1) {
          const uint32_t v = edge_col[i0 + u];
          qv[u] = load_pack<T, VEC>(Q + size_t(v) * G);
          if (!FIRST) rv[u] = load_pack<T, VEC>(R + size_t(i0 + u) * G);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u < e1) {
          const uint32_t slot = i0 + u - e0;
          const uint32_t v = edge_col[i0 + u];
          Pack<T, VEC> o, qn;
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            const T q = qv[u].v[k];
            const T r = FIRST ? T(0.0) : rv[u].v[k];
            const T x = q - r;
            const uint32_t neg = (x < T(0.0)) ? 1u : 0u;
            const T mag = (arg[k] == slot) ? min2[k] : min1[k];
            o.v[k] = (tot[k] ^ neg) ? -mag : mag;
            qn.v[k] = q + (o.v[k] - r);
          }
          T *rp = R + size_t(i0 + u) * G;
          T *qp = Q + size_t(v) * G;
          if (all_live) {
            store_pack<T, VEC>(rp, o);
            store_pack<T, VEC>(qp, qn);
          } else {
#pragma unroll
            for (int k = 0; k < VEC; k++)
              if (!frozen[k]) {
                rp[k] = o.v[k];
                qp[k] = qn.v[k];
              }
          }
        }
      }
    }
  }
}

// Register-resident form for levels whose rows have at most DMAX edges: the row's Qv and R
// values are loaded once and stay in VGPRs between the fold and the update (the update needs both
// originals: Qv + (out - R) in the reference's order), so HBM/L2 see 2 reads + 2 writes per edge
// instead of 4 + 2.  All of a row's loads are in flight together.  R is streamed (nontemporal).
template <typename T, int VEC, int DMAX, bool FIRST>
__global__ __launch_bounds__(256) void hl_minsum_reg_kernel(Graph g, Sched sc, State st,
                                                            const uint32_t *__restrict__ level_rows,
                                                            uint32_t n_level_rows, T *__restrict__ Q,
                                                            T *__restrict__ R) {
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const uint32_t waves_per_chunk = sc.waves_per_chunk;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * VEC;
  const size_t G = sc.tile;
  Q += tile_base(b0, g.n_cols, sc) + lane * VEC;
  R += tile_base(b0, g.n_edges, sc) + lane * VEC;
  bool frozen[VEC];
  bool any_live = false, all_live = true;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    frozen[k] = st.done[off + k] != 0;
    any_live = any_live || !frozen[k];
    all_live = all_live && !frozen[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;

  for (uint32_t idx = node0; idx < n_level_rows; idx += waves_per_chunk) {
    const uint32_t c = table_ptr(level_rows)[idx];
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    const uint32_t d = e1 - e0;
    if (d == 0) continue;
    uint32_t cols[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; i++) cols[i] = edge_col[e0 + min(uint32_t(i), d - 1)];
    Pack<T, VEC> q[DMAX], r[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
        q[i] = load_pack<T, VEC>(Q + size_t(cols[i]) * G);
        if (!FIRST) r[i] = load_msg<T, VEC, true>(R + size_t(e0 + i) * G);
      }
    }
    T min1[VEC], min2[VEC];
    uint32_t arg[VEC], tot[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      min1[k] = Limits<T>::inf();
      min2[k] = Limits<T>::inf();
      arg[k] = 0;
      tot[k] = 0;
    }
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const T x = FIRST ? (q[i].v[k] - T(0.0)) : (q[i].v[k] - r[i].v[k]);
          const T a = m_abs(x);
          if (x < T(0.0)) tot[k] ^= 1u;
          if (a < min1[k]) {
            min2[k] = min1[k];
            min1[k] = a;
            arg[k] = uint32_t(i);
          } else if (a < min2[k]) {
            min2[k] = a;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
        Pack<T, VEC> o, qn;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const T qq = q[i].v[k];
          const T rr = FIRST ? T(0.0) : r[i].v[k];
          const T x = qq - rr;
          const uint32_t neg = (x < T(0.0)) ? 1u : 0u;
          const T mag = (arg[k] == uint32_t(i)) ? min2[k] : min1[k];
          o.v[k] = (tot[k] ^ neg) ? -mag : mag;
          qn.v[k] = qq + (o.v[k] - rr);
        }
        T *rp = R + size_t(e0 + i) * G;
        T *qp = Q + size_t(cols[i]) * G;
        if (all_live) {
          store_msg<T, VEC, true>(rp, o);
          store_pack<T, VEC>(qp, qn);
        } else {
#pragma unroll
          for (int k = 0; k < VEC; k++)
            if (!frozen[k]) {
              rp[k] = o.v[k];
              qp[k] = qn.v[k];
            }
        }
      }
    }
  }
}

// Layered min-sum with ROW RECORDS (round 3): as in the flooding record kernel, a min-sum row's d messages R are the
// record {min1, min2, flip bits | argmin} (RowRec: R_i = (i == argmin ? min2 : min1) with sign bit flip[i], bit for bit
// the stored value), so the row reads and writes 3 (4) words instead of 2 d: per row 2 d + 6 words move where
// hl_minsum_reg_kernel moves 4 d (5G NR BG1: 0.72 of the traffic).  In the layered schedule a row touches only its own
// record: one buffer, updated in place; R of the first iteration is +0.0 (FIRST).  rec [M * RECW][tile] lives in the
// workspace's message array.
template <typename T, int VEC, int DMAX, int RECW, bool FIRST>
__global__ __launch_bounds__(256) void hl_minsum_rec_kernel(Graph g, Sched sc, State st,
                                                            const uint32_t *__restrict__ level_rows,
                                                            uint32_t n_level_rows, T *__restrict__ Q,
                                                            T *__restrict__ rec) {
  typedef typename RecWord<T>::type W;
  if (group_finished(st)) return;
  const TablePtr row_ptr = table_ptr(g.row_ptr);
  const TablePtr edge_col = table_ptr(g.edge_col);
  const uint32_t waves_per_chunk = sc.waves_per_chunk, tile = sc.tile;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  uint32_t chunk, node0;
  wave_slot(sc, wave, &chunk, &node0);
  if (chunk >= sc.nchunks) return;
  const uint32_t b0 = chunk * (64 * VEC);
  if (b0 >= *st.n_slots) return;
  const size_t off = size_t(b0) + lane * VEC;
  const size_t G = tile;
  Q += tile_base(b0, g.n_cols, sc) + lane * VEC;
  const uint32_t row_bytes = tile * uint32_t(sizeof(T)), lane_off = lane * uint32_t(VEC * sizeof(T));
  const RowBuf b_rec = row_buf(rec + tile_base(b0, g.n_rows * RECW, sc),
                               uint64_t(g.n_rows) * RECW * row_bytes - in_tile_of(b0, sc) * uint32_t(sizeof(T)));
  bool frozen[VEC];
  bool any_live = false, all_live = true;
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    frozen[k] = st.done[off + k] != 0;
    any_live = any_live || !frozen[k];
    all_live = all_live && !frozen[k];
  }
  if (__builtin_amdgcn_ballot_w64(any_live) == 0) return;
  all_live = __builtin_amdgcn_ballot_w64(!all_live) == 0;

  for (uint32_t idx = node0; idx < n_level_rows; idx += waves_per_chunk) {
    const uint32_t c = table_ptr(level_rows)[idx];
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    const uint32_t d = e1 - e0;
    if (d == 0) continue;
    uint32_t cols[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; i++) cols[i] = edge_col[e0 + min(uint32_t(i), d - 1)];
    Pack<T, VEC> q[DMAX];
    RowRec<T, VEC, RECW> old;
    if (!FIRST) old.load(b_rec, lane_off, c * RECW * row_bytes, row_bytes);
#pragma unroll
    for (int i = 0; i < DMAX; i++)
      if (uint32_t(i) < d) q[i] = load_pack<T, VEC>(Q + size_t(cols[i]) * G);
    T min1[VEC], min2[VEC];
    uint32_t arg[VEC];
    W sgn[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      min1[k] = Limits<T>::inf();
      min2[k] = Limits<T>::inf();
      arg[k] = 0;
      sgn[k] = 0;
    }
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const T rr = FIRST ? T(0.0) : old.value(uint32_t(i), k);
          const T x = q[i].v[k] - rr;
          const T a = m_abs(x);
          if (x < T(0.0)) sgn[k] |= W(1) << i;
          if (a < min1[k]) {
            min2[k] = min1[k];
            min1[k] = a;
            arg[k] = uint32_t(i);
          } else if (a < min2[k]) {
            min2[k] = a;
          }
        }
      }
    }
    RowRec<T, VEC, RECW> out;
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      const uint32_t tot = (sizeof(W) == 8 ? __popcll(sgn[k]) : __popc(uint32_t(sgn[k]))) & 1u;
      out.min1.v[k] = min1[k];
      out.min2.v[k] = min2[k];
      const W fl = tot ? ~sgn[k] : sgn[k];
      if constexpr (RECW == 4) {
        out.flip.v[k] = fl;
        out.arg.v[k] = W(arg[k]);
      } else {
        out.flip.v[k] = (fl & ((W(1) << RecWord<T>::kArgShift) - 1)) | (W(arg[k]) << RecWord<T>::kArgShift);
      }
    }
#pragma unroll
    for (int i = 0; i < DMAX; i++) {
      if (uint32_t(i) < d) {
        Pack<T, VEC> qn;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const T rr = FIRST ? T(0.0) : old.value(uint32_t(i), k);
          qn.v[k] = q[i].v[k] + (out.value(uint32_t(i), k) - rr);  // Qv += out - R (arithmetic.rs:570-573 without the correction)
        }
        T *qp = Q + size_t(cols[i]) * G;
        if (all_live) {
          store_pack<T, VEC>(qp, qn);
        } else {
#pragma unroll
          for (int k = 0; k < VEC; k++)
            if (!frozen[k]) qp[k] = qn.v[k];
        }
      }
    }
    if (all_live) {
      out.template store<false>(b_rec, lane_off, c * RECW * row_bytes, row_bytes);
    } else {
      // a frozen codeword keeps its record (nothing reads it again, but nothing may be half-written either)
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        if (!frozen[k]) {
          const uint32_t lo = lane_off + k * uint32_t(sizeof(T));
          row_store<T, false>(b_rec, lo, c * RECW * row_bytes, out.min1.v[k]);
          row_store<T, false>(b_rec, lo, c * RECW * row_bytes + row_bytes, out.min2.v[k]);
          row_store<T, false>(b_rec, lo, c * RECW * row_bytes + 2 * row_bytes, __builtin_bit_cast(T, out.flip.v[k]));
          if constexpr (RECW == 4) row_store<T, false>(b_rec, lo, c * RECW * row_bytes + 3 * row_bytes, __builtin_bit_cast(T, out.arg.v[k]));
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Bookkeeping kernels
// ---------------------------------------------------------------------------------------
__global__ void init_group_kernel(uint32_t *done, int32_t *iters, uint32_t *unsat0, uint32_t *unsat1,
                                  uint32_t *n_active, uint32_t *n_slots, uint32_t *slot_cw, uint32_t nb,
                                  uint32_t G) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < G) {
    done[b] = b >= nb ? 1u : 0u;
    iters[b] = -1;
    unsat0[b] = 0;
    unsat1[b] = 0;
    slot_cw[b] = b < nb ? b : kNoCodeword;
  }
  if (b == 0) {
    *n_active = nb;
    *n_slots = min(G, (nb + 255u) / 256u * 256u);
  }
}

// A codeword whose syndrome flag stayed clear is finished at `iteration`
// (flooding.rs:57-64, 69-79; horizontal_layered.rs:55-62, 66-78).
__global__ void latch_kernel(uint32_t *done, int32_t *iters, uint32_t *unsat, uint32_t *n_active,
                             int32_t iteration, uint32_t G) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= G) return;
  if (!done[b] && unsat[b] == 0) {
    done[b] = 1u;
    iters[b] = iteration;
    atomicSub(n_active, 1u);
  }
  unsat[b] = 0;
}

// hard decisions (x <= 0, arithmetic.rs:198-200) of [N][G] soft values, bit-packed
// 64 codewords per word with a wave ballot: bits[v][w], W = G / 64 words per variable
template <typename T>
__global__ void pack_hard_kernel(const T *__restrict__ soft, uint64_t *__restrict__ bits,
                                 const uint32_t *__restrict__ n_active, const uint32_t *__restrict__ n_slots,
                                 uint32_t n_cols, uint32_t tile, uint32_t W, uint32_t waves_per_word) {
  if (*n_active == 0) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t w = wave / waves_per_word;
  if (w >= W || w * 64 >= *n_slots) return;
  soft += tile_base(w * 64, n_cols, tile) + lane;
  // eight rows in flight per wave: one 256-byte row at a time left the kernel latency-bound
  constexpr int U = 8;
  for (uint32_t v0 = wave % waves_per_word; v0 < n_cols; v0 += U * waves_per_word) {
    T x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t v = v0 + u * waves_per_word;
      if (v < n_cols) x[u] = soft[size_t(v) * tile];  // wave-uniform guard
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t v = v0 + u * waves_per_word;
      if (v < n_cols) {
        const uint64_t b = __builtin_amdgcn_ballot_w64(x[u] <= T(0.0));
        if (lane == 0) bits[size_t(v) * W + w] = b;
      }
    }
  }
}

// the same with paired loads: a lane loads two neighbouring codewords (4, 8 or 16 bytes per lane instead of 2, 4
// or 8: a wavefront's request covers 128 codewords), forms their two decisions, and lane i then fetches codeword
// i's decision from lane i / 2 (ds_bpermute) for the first packed word and from lane 32 + i / 2 for the second.
// Needs tiles of a multiple of 128 codewords.
template <typename T>
__global__ void pack_hard_pair_kernel(const T *__restrict__ soft, uint64_t *__restrict__ bits,
                                      const uint32_t *__restrict__ n_active, const uint32_t *__restrict__ n_slots,
                                      uint32_t n_cols, uint32_t tile, uint32_t W, uint32_t waves_per_pair) {
  if (*n_active == 0) return;
  struct alignas(2 * sizeof(T)) Two {
    T a, b;
  };
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t w = (wave / waves_per_pair) * 2;  // packed words w and w + 1: codewords [64 w, 64 w + 128)
  if (w >= W || w * 64 >= *n_slots) return;
  const Two *__restrict__ src = reinterpret_cast<const Two *>(soft + tile_base(w * 64, n_cols, tile)) + lane;
  const uint32_t row_pairs = tile / 2;
  const int from0 = static_cast<int>((lane >> 1) * 4), from1 = static_cast<int>((32 + (lane >> 1)) * 4);
  const uint32_t which = lane & 1u;
  constexpr int U = 8;
  for (uint32_t v0 = wave % waves_per_pair; v0 < n_cols; v0 += U * waves_per_pair) {
    Two x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t v = v0 + u * waves_per_pair;
      if (v < n_cols) x[u] = src[size_t(v) * row_pairs];  // wave-uniform guard
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t v = v0 + u * waves_per_pair;
      if (v < n_cols) {
        const int two = (x[u].a <= T(0) ? 1 : 0) | (x[u].b <= T(0) ? 2 : 0);  // arithmetic.rs:198-200
        const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(from0, two));
        const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(from1, two));
        const uint64_t b0 = __builtin_amdgcn_ballot_w64(((lo >> which) & 1u) != 0);
        const uint64_t b1 = __builtin_amdgcn_ballot_w64(((hi >> which) & 1u) != 0);
        if (lane == 0) {
          bits[size_t(v) * W + w] = b0;
          if (w + 1 < W) bits[size_t(v) * W + w + 1] = b1;
        }
      }
    }
  }
}

// syndrome of packed hard decisions (decoder.rs:157-164): wavefront = (block of checks, 64 packed words),
// lane = word; sets unsat[b] = 1 for every codeword with at least one odd check.  The checks and their
// variable lists are wave-uniform (scalar loads, eight indices ahead), the eight 512-byte reads of a step are
// in flight together.
__global__ __launch_bounds__(256) void syndrome_bits_kernel(const uint32_t *__restrict__ row_ptr_,
                                                            const uint32_t *__restrict__ edge_col_, uint32_t n_rows,
                                                            const uint64_t *__restrict__ bits,
                                                            uint32_t *__restrict__ unsat,
                                                            const uint32_t *__restrict__ n_active,
                                                            const uint32_t *__restrict__ n_slots, uint32_t W,
                                                            uint32_t rows_per_wave) {
  if (*n_active == 0) return;
  constexpr int U = 8;
  const TablePtr row_ptr = table_ptr(row_ptr_), edge_col = table_ptr(edge_col_);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t word_chunks = (W + 63) / 64;
  const uint32_t w = (wave % word_chunks) * 64 + lane;
  const uint32_t c0 = (wave / word_chunks) * rows_per_wave;
  if (c0 >= n_rows) return;
  const bool live = w < W && w * 64 < *n_slots;
  if (__builtin_amdgcn_ballot_w64(live) == 0) return;
  const uint32_t c1 = min(c0 + rows_per_wave, n_rows);
  bits += live ? w : 0;
  uint64_t acc = 0;
  for (uint32_t c = c0; c < c1; c++) {
    const uint32_t e0 = row_ptr[c], e1 = row_ptr[c + 1];
    uint64_t x = 0;
    for (uint32_t e = e0; e < e1; e += U) {
      uint64_t y[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t v = edge_col[min(e + u, e1 - 1)];
        y[u] = bits[size_t(v) * W];
      }
#pragma unroll
      for (int u = 0; u < U; u++) x ^= (e + u < e1) ? y[u] : 0;
    }
    acc |= x;
  }
  if (!live) acc = 0;
  // flags of the codewords with an odd check: one packed word at a time, its set bits as the lane mask of one
  // coalesced store (lane = bit).  (A lane walking the set bits of its own word issued up to 64 scattered stores
  // per lane -- 36 M stores per launch when no codeword of 8192 has converged, 100 us of a 140 us kernel.)
  const uint32_t w0 = (wave % word_chunks) * 64;
  for (uint32_t j = 0; j < 64; j++) {
    const uint64_t bitsj = (uint64_t(uint32_t(__builtin_amdgcn_readlane(static_cast<int>(acc >> 32), j))) << 32) |
                           uint64_t(uint32_t(__builtin_amdgcn_readlane(static_cast<int>(acc), j)));
    if (bitsj == 0) continue;  // wave-uniform (also: lanes that are not live carry acc = 0)
    if ((bitsj >> lane) & 1ull) unsat[size_t(w0 + j) * 64 + lane] = 1u;
  }
}

// ---------------------------------------------------------------------------------------
// Layout changes at the boundary: callers hand over codeword-major rows
// ([batch][len], the layout of a loop of scalar decode calls), the kernels work on
// [node][G].  64x64 tiles through LDS, both sides coalesced.
// ---------------------------------------------------------------------------------------

// Reads the caller's LLR rows, depunctures (puncturing.rs:83-101: punctured blocks become
// 0.0 LLRs), quantises to the arithmetic type (`x as f32`, arithmetic.rs:194-196), writes
// chan and post (= L_0), and packs the hard decisions of the RAW input for the pre-check
// (flooding.rs:57).  Lanes beyond the batch are padded with +1.0.
template <typename SrcT, typename T>
__global__ __launch_bounds__(256) void ingest_kernel(const SrcT *__restrict__ src, size_t src_stride,
                                                     uint32_t nb, uint32_t n, uint32_t G, uint32_t tile,
                                                     T *__restrict__ chan, T *__restrict__ post,
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
    if (v < n) {  // wave-uniform
      const SrcT val = lds[tx][r];
      const T q = static_cast<T>(val);
      chan[base + size_t(v) * tile] = q;
      post[base + size_t(v) * tile] = q;
      const uint64_t bal = __builtin_amdgcn_ballot_w64(val <= SrcT(0.0));
      if (tx == 0) rawbits[size_t(v) * W + blockIdx.y] = bal;
    }
  }
}

// post -> the caller's rows: bits [batch][out_len] u8, iterations [batch], posterior [batch][n]
// (optional).  Slot s of the group holds codeword slot_cw[s].  retire_only: write just the
// finished codewords (called right before a compaction drops them from the group), and only if
// the compaction was decided (*do_compact).  Codewords that passed the pre-check report the
// hard decisions of the raw input (flooding.rs:59-63).  zero_fill: the reference's
// max_iterations = 0 failure of the flooding decoder reports its never-written output_llrs
// (flooding.rs:27-28, 82-85).
template <typename T, typename OutT>
__global__ __launch_bounds__(256) void emit_kernel(const T *__restrict__ post,
                                                   const uint64_t *__restrict__ rawbits, State st,
                                                   const uint32_t *__restrict__ do_compact, uint32_t n,
                                                   uint32_t G, uint32_t tile, uint32_t out_len,
                                                   uint8_t *__restrict__ bits, int32_t *__restrict__ iterations,
                                                   OutT *__restrict__ posterior, int zero_fill,
                                                   int retire_only) {
  __shared__ T lds[64][65];
  const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
  const uint32_t b0 = blockIdx.y * 64;
  if (b0 >= *st.n_slots) return;
  if (retire_only && *do_compact == 0) return;
  if (retire_only) {  // nothing to retire among this block's 64 slots (wave-uniform: every wave looks at the same 64)
    const uint32_t slot = b0 + tx;
    if (__builtin_amdgcn_ballot_w64(st.done[slot] != 0 && st.slot_cw[slot] != kNoCodeword) == 0) return;
  }
  const size_t base = tile_base(b0, n, tile) + tx;
  const uint32_t W = G / 64;
  // only the rows somebody asked for: the first out_len hard decisions, all n soft values if a posterior is wanted
  const uint32_t n_emit = posterior ? n : min(n, out_len);
  for (uint32_t v0 = blockIdx.x * 64; v0 < max(n_emit, 1u); v0 += gridDim.x * 64) {
  __syncthreads();
  for (uint32_t r = ty; r < 64; r += 4) {
    const uint32_t v = v0 + r;
    lds[r][tx] = (v < n) ? post[base + size_t(v) * tile] : T(0.0);
  }
  __syncthreads();
  for (uint32_t r = ty; r < 64; r += 4) {
    const uint32_t slot = b0 + r, v = v0 + tx;
    const uint32_t cw = st.slot_cw[slot];
    if (cw == kNoCodeword) continue;                       // wave-uniform
    if (retire_only && st.done[slot] == 0) continue;       // wave-uniform
    const int32_t it = st.iters[slot];
    if (v < n) {
      T val = lds[tx][r];
      uint8_t bit;
      if (it == 0 && rawbits != nullptr)  // (continuous batching passes none: its f32 inputs are their own quantisation)
        bit = uint8_t((rawbits[size_t(v) * W + (cw >> 6)] >> (cw & 63u)) & 1u);
      else if (zero_fill && it < 0) {
        bit = 1;
        val = T(0.0);
      } else
        bit = uint8_t(val <= T(0.0));
      if (v < out_len) bits[size_t(cw) * out_len + v] = bit;
      if constexpr (sizeof(T) == 2) {
        // i8 arithmetics: the soft output is the 8-bit LLR clip(llr) (arithmetic.rs:651, 713-715)
        const int c = val >= 127 ? 127 : (val <= -127 ? -127 : int(val));
        if (posterior) posterior[size_t(cw) * n + v] = static_cast<OutT>(c);
      } else {
        if (posterior) posterior[size_t(cw) * n + v] = static_cast<OutT>(val);
      }
    }
    if (iterations && v0 == 0 && tx == 0) iterations[cw] = it;
  }
  }
}

// ---------------------------------------------------------------------------------------
// The syndrome test as an operator (decoder.rs:157-164 keeps only "is it zero?"; here the
// parities themselves are returned): hard decisions in the callers' layout, bits [batch][n] one
// byte per bit -> syndrome [batch][m] (1 = unsatisfied check, optional) and weight [batch]
// (optional).  A thread owns one (codeword, check); a wave's 64 checks are consecutive rows of one
// codeword, whose 64 KB of bits stay in L2.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void syndrome_of_bits_kernel(const uint32_t *__restrict__ row_ptr,
                                                               const uint32_t *__restrict__ edge_col,
                                                               uint32_t m, uint32_t n, uint32_t batch,
                                                               const uint8_t *__restrict__ bits,
                                                               uint8_t *__restrict__ syndrome,
                                                               uint32_t *__restrict__ weight) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = blockIdx.y;
  if (b >= batch) return;
  uint32_t parity = 0;
  if (c < m) {
    const uint8_t *row = bits + size_t(b) * n;
    for (uint32_t e = row_ptr[c]; e < row_ptr[c + 1]; e++) parity ^= row[edge_col[e]] & 1u;
    if (syndrome) syndrome[size_t(b) * m + c] = static_cast<uint8_t>(parity);
  }
  if (weight) {
    const uint64_t odd = __builtin_amdgcn_ballot_w64(parity != 0);
    if ((threadIdx.x & 63u) == 0 && odd != 0) atomicAdd(weight + b, static_cast<uint32_t>(__popcll(odd)));
  }
}

// ---------------------------------------------------------------------------------------
// Batch compaction.  With syndrome early termination the finished codewords of a group stop
// being rewritten but their slots still cost a pass of every kernel until the whole 256-wide
// tile is finished.  At a checkpoint the live codewords are packed into the leading slots:
//   plan     counts the live codewords, decides whether packing pays; the live codewords beyond the
//            new end of the group ("movers") are paired, in order, with the finished slots below it ("holes")
//   emit     (retire_only) writes the results of the finished codewords to the caller
//   move     array[row][hole_i] = array[row][mover_i]      for every state array (sources and destinations
//            are disjoint: one pass, no staging; a live codeword that is already below the new end stays put)
//   commit   new flags, slot_cw, n_slots
// Everything is decided on the device (no host synchronisation); when packing does not pay the
// kernels return at once.  (Round 1 moved every live codeword through a staging copy -- a stable
// partition: twice the traffic for all of them instead of once for the movers, 10 % of a 2 dB batch.)
// ---------------------------------------------------------------------------------------
struct CompactPlan {
  uint32_t do_compact;  // decided by compact_plan_kernel
  uint32_t n_live;      // live codewords
  uint32_t new_slots;   // n_live rounded up to 256
  uint32_t n_move;      // live codewords at or beyond new_slots = holes that get filled
};

// one workgroup of 1024 threads; G <= 64 K slots
struct CompactRule {
  uint32_t horizon;      // iterations a freed slot is assumed to save at most
  uint32_t cost_live;    // cost of the move per live codeword, in quarter codeword-iterations
  uint32_t cost_slots;   // ... and per slot of the group before the move
  uint32_t min_freed_q;  // at least this many quarters of the slots must be freed
};

// movers[i] / holes[i]: slot pairs of the move; fill_cw[s]: the codeword that lands in slot s (kNoCodeword: none)
__global__ __launch_bounds__(1024) void compact_plan_kernel(State st, CompactPlan *plan, uint32_t *movers,
                                                           uint32_t *holes, uint32_t *fill_cw,
                                                           uint32_t remaining_iterations, CompactRule rule) {
  __shared__ uint32_t wave_tot[2][16];
  __shared__ uint32_t base[2];
  __shared__ uint32_t s_new_slots, s_go;
  // the checkpoints also publish the progress word for the schedules whose check-node kernels do not
  // (the streaming flooding kernels): the host stops enqueuing a finished group at the next one
  if (threadIdx.x == 0 && st.publish != nullptr)
    __hip_atomic_store(st.publish, progress_word(st.epoch, st.tick, min(*st.n_active, 0xFFFFFu)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  const uint32_t n_slots = *st.n_slots;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) base[0] = base[1] = 0;
  __syncthreads();
  // pass 1: how many are live
  uint32_t mine = 0;
  for (uint32_t s = threadIdx.x; s < n_slots; s += 1024) mine += st.done[s] == 0 ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if (lane == 0) wave_tot[0][wid] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t n_live = 0;
    for (uint32_t i = 0; i < 16; i++) n_live += wave_tot[0][i];
    const uint32_t new_slots = (n_live + 255u) / 256u * 256u;
    plan->n_live = n_live;
    plan->new_slots = new_slots;
    plan->n_move = 0;
    // packing saves the freed slots' share of the iterations still to come -- of which only a handful
    // are likely (the group is converging), so the horizon is capped; a minimum share of the slots must
    // be freed, or successive checkpoints would keep re-packing for crumbs (measured,
    // tools/compaction_sweep.py)
    const uint32_t freed = n_slots - min(new_slots, n_slots);
    const uint64_t gain = uint64_t(freed) * min(remaining_iterations, rule.horizon) * 4;
    const uint64_t cost = uint64_t(n_live) * rule.cost_live + uint64_t(n_slots) * rule.cost_slots;
    const uint32_t go =
        (n_live > 0 && uint64_t(freed) * 4 >= uint64_t(n_slots) * rule.min_freed_q && freed > 0 && gain > cost) ? 1u : 0u;
    plan->do_compact = go;
    s_go = go;
    s_new_slots = new_slots;
  }
  __syncthreads();
  if (!s_go) return;
  const uint32_t new_slots = s_new_slots;
  // pass 2: the movers and the holes, each in slot order
  for (uint32_t s0 = 0; s0 < n_slots; s0 += 1024) {
    const uint32_t s = s0 + threadIdx.x;
    const bool in = s < n_slots;
    const bool live = in && st.done[s] == 0;
    const bool cls[2] = {live && s >= new_slots, in && !live && s < new_slots};
    uint32_t before[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint64_t m = __builtin_amdgcn_ballot_w64(cls[q]);
      before[q] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wave_tot[q][wid] = __popcll(m);
    }
    if (in) fill_cw[s] = kNoCodeword;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++) {
      uint32_t off = base[q];
      for (uint32_t i = 0; i < wid; i++) off += wave_tot[q][i];
      if (cls[q]) (q == 0 ? movers : holes)[off + before[q]] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      uint32_t t = 0;
      for (uint32_t i = 0; i < 16; i++) t += wave_tot[threadIdx.x][i];
      base[threadIdx.x] += t;
    }
    __syncthreads();
  }
  // pass 3: who lands where (there are at least as many holes as movers: new_slots >= n_live)
  const uint32_t n_move = base[0];
  for (uint32_t i = threadIdx.x; i < n_move; i += 1024) fill_cw[holes[i]] = st.slot_cw[movers[i]];
  if (threadIdx.x == 0) plan->n_move = n_move;
}

// The state arrays moved by a compaction: (pointer, rows) x count
template <typename T>
struct MoveList {
  T *arr[3];
  uint32_t rows[3];   // rows of the array (its tile stride)
  uint32_t moved[3];  // the leading rows that travel (<= rows)
  uint32_t count;
};

// arr[row][holes[i]] = arr[row][movers[i]]: a wavefront takes 64 pairs and every waves_per_chunk-th row,
// eight rows in flight
template <typename T>
__global__ __launch_bounds__(256) void compact_move_kernel(const CompactPlan *plan,
                                                           const uint32_t *__restrict__ movers,
                                                           const uint32_t *__restrict__ holes, MoveList<T> ml,
                                                           uint32_t tile, uint32_t nchunks, uint32_t waves_per_chunk) {
  if (plan->do_compact == 0) return;
  constexpr int U = 8;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const uint32_t chunk = wave / waves_per_chunk;
  if (chunk >= nchunks || chunk * 64 >= plan->n_move) return;
  const uint32_t i = chunk * 64 + lane;
  const bool valid = i < plan->n_move;
  const uint32_t from = valid ? movers[i] : 0, to = valid ? holes[i] : 0;
  const uint32_t r0 = wave % waves_per_chunk;
  for (uint32_t a = 0; a < ml.count; a++) {
    const uint32_t rows = ml.rows[a];
    const T *__restrict__ src = ml.arr[a] + tile_base(from, rows, tile);
    T *__restrict__ dst = ml.arr[a] + tile_base(to, rows, tile);
    const uint32_t moved = ml.moved[a];
    for (uint32_t r = r0; r < moved; r += U * waves_per_chunk) {
      T x[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t ru = r + u * waves_per_chunk;
        if (valid && ru < moved) x[u] = src[size_t(ru) * tile];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t ru = r + u * waves_per_chunk;
        if (valid && ru < moved) dst[size_t(ru) * tile] = x[u];
      }
    }
  }
}

__global__ void compact_commit_kernel(State st, const CompactPlan *plan, uint32_t *unsat0, uint32_t *unsat1,
                                      uint32_t *n_slots_w, const uint32_t *__restrict__ fill_cw, uint32_t G) {
  if (plan->do_compact == 0) return;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= G) return;
  // codewords change slices: from here on every slice stores its L-free posteriors (State::slice_state)
  if (st.slice_state != nullptr && b < G / 64) st.slice_state[b] = 2;
  if (b >= plan->new_slots) {
    st.slot_cw[b] = kNoCodeword;
    st.done[b] = 1u;
  } else if (st.done[b] != 0) {
    const uint32_t cw = fill_cw[b];  // (new_slots <= the old group size: every slot below it was classified)
    st.slot_cw[b] = cw;
    st.done[b] = cw == kNoCodeword ? 1u : 0u;
  }
  st.iters[b] = -1;
  unsat0[b] = 0;
  unsat1[b] = 0;
  if (b == 0) *n_slots_w = plan->new_slots;
}

// ---------------------------------------------------------------------------------------
// Continuous batching (DeviceDecoder::decode_stream; the reference's workers produce frames until the stop rule
// fires, /root/reference/src/simulation/ber.rs:297-368, 522-531).  The group never drains: every `harvest`
//   emit    (retire_only) writes the results of the finished codewords to the caller's rows
//   plan    lists the free slots (finished codewords and slots never filled) and hands the next codewords of
//           the stream to them, as many as are left; publishes the progress for the host
//   source  (the caller's kernels) produces those codewords' LLR rows in a staging buffer
//   ingest  moves the rows into the freed slots' columns of chan / post and restarts the slots' state
// A refilled slot needs no other preparation: its first check-node pass reads no messages (STREAM).
// ---------------------------------------------------------------------------------------
struct StreamPlan {
  uint64_t first;      // index of the first codeword handed out by this harvest (what the source kernels read, with count)
  uint64_t count;      // codewords handed out by this harvest
  uint64_t next;       // codewords handed out so far
  uint64_t retired;    // codewords whose results have been written
  uint64_t total;      // codewords of the stream
  uint32_t always;     // = 1: the flag emit_kernel's retire mode looks at
  uint32_t pad;
};

#ifdef LDPC_EXPERIMENTS  // continuous batching is exact and loses to drained batches with compaction (round 3): not in the product
// one workgroup of 1024 threads; G <= 64 K slots.  holes[i] = i-th free slot (slot order); the first `count` get
// codewords first + i.  progress: pinned host word <- (epoch << 40) | retired (the host stops when retired == total).
__global__ __launch_bounds__(1024) void stream_plan_kernel(State st, StreamPlan *plan, uint32_t *holes, uint32_t G,
                                                          uint64_t *progress, uint32_t epoch) {
  __shared__ uint32_t wave_tot[2][16];
  __shared__ uint32_t base[2];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) base[0] = base[1] = 0;
  __syncthreads();
  for (uint32_t s0 = 0; s0 < G; s0 += 1024) {
    const uint32_t s = s0 + threadIdx.x;
    const bool in = s < G;
    const bool hole = in && st.done[s] != 0;
    const bool finished = hole && st.slot_cw[s] != kNoCodeword;  // emitted by the retire pass just before this kernel
    const bool cls[2] = {hole, finished};
    uint32_t before[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint64_t m = __builtin_amdgcn_ballot_w64(cls[q]);
      before[q] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wave_tot[q][wid] = __popcll(m);
    }
    if (finished) st.slot_cw[s] = kNoCodeword;  // never emitted twice
    __syncthreads();
    uint32_t off = base[0];
    for (uint32_t i = 0; i < wid; i++) off += wave_tot[0][i];
    if (hole) holes[off + before[0]] = s;
    __syncthreads();
    if (threadIdx.x < 2) {
      uint32_t t = 0;
      for (uint32_t i = 0; i < 16; i++) t += wave_tot[threadIdx.x][i];
      base[threadIdx.x] += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint64_t left = plan->total - plan->next;
    const uint64_t count = left < base[0] ? left : base[0];
    plan->first = plan->next;
    plan->count = count;
    plan->next += count;
    plan->retired += base[1];
    plan->always = 1;
    __hip_atomic_store(progress, (uint64_t(epoch & 0xFFFFFFu) << 40) | (plan->retired & 0xFFFFFFFFFFull), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// staging [count][src_stride] rows -> the columns of chan / post at slots holes[0 .. count); restarts those slots
// (done, iteration count, start iteration, row in the caller's arrays).  Depuncture and quantisation as ingest_kernel.
// grid (ceil(n / 64), ceil(G / 64)): block (x, y) moves variables [64x, 64x + 64) of holes [64y, 64y + 64).
template <typename SrcT, typename T>
__global__ __launch_bounds__(256) void stream_ingest_kernel(const SrcT *__restrict__ src, size_t src_stride,
                                                            const StreamPlan *__restrict__ plan,
                                                            const uint32_t *__restrict__ holes, State st, uint32_t *it0,
                                                            uint32_t now, uint32_t n, uint32_t tile, T *__restrict__ chan,
                                                            T *__restrict__ post, uint32_t *__restrict__ unsat0,
                                                            uint32_t *__restrict__ unsat1,
                                                            const int32_t *__restrict__ src_block, uint32_t block_size) {
  __shared__ SrcT lds[64][65];
  __shared__ uint32_t s_slot[64];
  const uint32_t count = static_cast<uint32_t>(plan->count);
  const uint32_t h0 = blockIdx.y * 64;
  if (h0 >= count) return;
  const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
  const uint32_t v0 = blockIdx.x * 64;
  if (threadIdx.x < 64) s_slot[threadIdx.x] = h0 + threadIdx.x < count ? holes[h0 + threadIdx.x] : kNoCodeword;
  for (uint32_t r = ty; r < 64; r += 4) {
    const uint32_t h = h0 + r, v = v0 + tx;
    SrcT val = SrcT(1.0);
    if (h < count && v < n) {
      if (src_block) {
        const int32_t sb = src_block[v / block_size];
        val = sb < 0 ? SrcT(0.0) : src[size_t(h) * src_stride + size_t(sb) * block_size + v % block_size];
      } else {
        val = src[size_t(h) * src_stride + v];
      }
    }
    lds[r][tx] = val;
  }
  __syncthreads();
  const uint32_t slot = s_slot[tx];
  if (slot != kNoCodeword) {
    const size_t base = (size_t(slot / tile) * n) * tile + slot % tile;
    for (uint32_t r = ty; r < 64; r += 4) {
      const uint32_t v = v0 + r;
      if (v < n) {
        const T q = static_cast<T>(lds[tx][r]);
        chan[base + size_t(v) * tile] = q;
        post[base + size_t(v) * tile] = q;
      }
    }
    if (blockIdx.x == 0 && ty == 0) {
      st.done[slot] = 0u;
      st.iters[slot] = -1;
      st.slot_cw[slot] = static_cast<uint32_t>(plan->first) + h0 + tx;
      it0[slot] = now;
      unsat0[slot] = 0u;
      unsat1[slot] = 0u;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(st.n_active, count);
}
#endif  // LDPC_EXPERIMENTS (continuous batching)

}  // namespace dev
}  // namespace ldpc
