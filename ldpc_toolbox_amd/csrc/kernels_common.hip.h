// Shared by every kernel of the batched decoder: the tiled layout and its buffer descriptors, Graph / Sched / State, the
// progress word, the exact-math wrappers and the check-node rules in the reference's operation order (rule_check_node).
// Part of kernels.hip.h (include that).
#pragma once
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
// the fused forms (exact_math.h): the same operations per lane, one straight line instead of three functions' branches
template <>
__device__ __forceinline__ float phi_fn<float>(float x) {
  return em::phif(x);
}
#ifndef LDPC_GENERIC_PHI64  // (A/B builds: -DLDPC_GENERIC_PHI64 keeps -log(tanh(.)) as three calls, round 5's form)
template <>
__device__ __forceinline__ double phi_fn<double>(double x) {
  return em::phi(x);
}
#endif
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

}  // namespace dev
}  // namespace ldpc
