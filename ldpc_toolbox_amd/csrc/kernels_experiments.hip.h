// Opt-in forms that measured level with or behind the default paths and left the product in round 5: the slice-persistent
// layered kernel (one launch per iteration, round 4) and continuous batching's plan / ingest kernels (round 3).  Compiled only
// with -DLDPC_EXPERIMENTS (tools/ab_variants.sh build exp -DLDPC_EXPERIMENTS); their tests run on such a build.
// Part of kernels.hip.h (include that).
#pragma once
#ifdef LDPC_EXPERIMENTS
namespace ldpc {
namespace dev {

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

// one workgroup of 1024 threads; G <= 64 K slots.  holes[i] = i-th free slot (slot order); the first `count` get
// codewords first + i.  progress: pinned host word <- (epoch << 40) | retired (the host stops when retired == total).
#ifdef LDPC_STREAM_KERNELS_TU  // not a template: compiled in the translation unit of decode_stream (run_group_f32.hip)
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
#endif  // LDPC_STREAM_KERNELS_TU

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

}  // namespace dev
}  // namespace ldpc
#endif  // LDPC_EXPERIMENTS
