// Layered schedule: one launch per dependency level (two-pass and register-resident rows), the flooding Tanh rule's
// register-resident rows (cn_reg_kernel: the same row handling), layered min-sum (streaming, register-resident, row records).
// Part of kernels.hip.h (include that).
#pragma once
namespace ldpc {
namespace dev {

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

}  // namespace dev
}  // namespace ldpc
