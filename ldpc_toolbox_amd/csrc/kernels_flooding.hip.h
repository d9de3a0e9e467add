// Flooding schedule: min-sum check nodes (per-edge messages, L-free variant, ROW RECORDS -- the headline kernel), the
// LDS-staged check nodes of the other rules, the variable-node kernel.  Part of kernels.hip.h (include that).
#pragma once
namespace ldpc {
namespace dev {

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
// EVW != 0 (round 5; flooding min-sum with row records and deferred L-free stores): the launch also does what a separate
// vn_free_rec_kernel launch did after it in every iteration -- the L-free posteriors of the FIRST codewords of a slice to
// converge, rebuilt from the records of the iteration being latched (State::slice_state) -- spread over the launch's own
// waves instead of a small grid of its own: one launch and one dispatch gap fewer per iteration (4.4 + 5.7 us of 2050), and
// the one pass that does find work runs at the full grid's width.  Which codewords are "newly converged" must not depend on
// what the bookkeeping wave of the slice has already written in this same launch: see `fresh` below.
template <typename T>
struct VnEvent {
  const uint32_t *free_var, *free_rs;  // the L-free variables and, per variable, its two (row << 6 | slot) words
  const T *rec;                        // records of the iteration being latched
  uint32_t n_free;
};
template <typename T, int VEC, int U, bool NT, bool LIST, int EVW = 0>
__global__ __launch_bounds__(256) void vn_kernel(
    Graph g, Sched sc, State st, const T *__restrict__ chan, const T *__restrict__ msg,
    T *__restrict__ post, const uint32_t *__restrict__ unsat_in, uint32_t *__restrict__ unsat_clear,
    int32_t latch_iteration, VnEvent<T> ev = VnEvent<T>{nullptr, nullptr, nullptr, 0}) {
  uint32_t *__restrict__ n_active = st.n_active;
  // A finished group's launches return at once.  With EVW the count can also reach zero INSIDE this launch -- the
  // bookkeeping waves below subtract the codewords they latch -- and a wave that starts after the last subtraction must
  // still take its share of the rebuild of those codewords' L-free posteriors: it goes on to the EVW block (where a group
  // that had finished BEFORE the launch has no `fresh` codeword and costs a few loads) and returns behind it.
  const bool idle = *n_active == 0;
  if (EVW == 0 && idle) return;
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
  if constexpr (EVW != 0) {
    if (idle && (ev.rec == nullptr || unsat_in == nullptr || st.slice_state == nullptr || st.slice_state[chunk] == 2u)) return;
  }
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
    if (v_first == 0 && !idle) {
      // exactly one wave per slice does the per-codeword bookkeeping (idle: the count is zero only once EVERY bookkeeping
      // wave has subtracted its codewords, this one included -- or the group had finished before the launch)
      if (converged || expired) {
        done[off + k] = 1u;
        iters[off + k] = converged ? own_iterations : -1;
        atomicSub(n_active, 1u);
        any_new = true;
      }
      unsat_clear[off + k] = 0u;
    }
  }
  if constexpr (EVW != 0) {
    // The slice's first convergences (slice_state 0, or 1 when the bookkeeping wave has already marked it in this launch;
    // 2 = the check-node kernel has been storing the L-free posteriors all along): every wave of the slice takes its share
    // of the L-free variables.  `fresh`: converging in THIS launch -- from what this launch does not change (the syndrome
    // flag the last check-node pass left, the slot's codeword) and from `iters`, which is -1 before the launch and
    // latch_iteration once the bookkeeping wave has been here: both mean "this launch" (an earlier convergence carries its
    // own, smaller count; an empty slot has no codeword).
    if (ev.rec != nullptr && unsat_in != nullptr && st.slice_state != nullptr && st.slice_state[chunk] != 2u) {
      bool fresh[VEC];
      bool any_fresh = false;
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const int32_t was = iters[off + k];
        fresh[k] = unsat_in[off + k] == 0 && st.slot_cw[off + k] != kNoCodeword && (was < 0 || was == latch_iteration);
        any_fresh = any_fresh || fresh[k];
      }
      if (__builtin_amdgcn_ballot_w64(any_fresh) != 0) {
        const TablePtr free_var = table_ptr(ev.free_var), free_rs = table_ptr(ev.free_rs);
        const uint32_t row_bytes = tile * uint32_t(sizeof(T)), lane_off = lane * uint32_t(VEC * sizeof(T));
        const RowBuf b_rec = row_buf(ev.rec + tile_base(b0, g.n_rows * EVW, sc),
                                     uint64_t(g.n_rows) * EVW * row_bytes - in_tile_of(b0, sc) * uint32_t(sizeof(T)));
        for (uint32_t i = v_first; i < ev.n_free; i += waves_per_chunk) {
          const uint32_t fv = free_var[i], a = free_rs[2 * i], b = free_rs[2 * i + 1];
          const Pack<T, VEC> ch = load_pack<T, VEC>(chan + size_t(fv) * G);
          RowRec<T, VEC, EVW> ra, rb;
          if (a != kAuxNone) ra.load(b_rec, lane_off, (a >> 6) * EVW * row_bytes, row_bytes);
          if (b != kAuxNone) rb.load(b_rec, lane_off, (b >> 6) * EVW * row_bytes, row_bytes);
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            T sum = -T(0.0);  // arithmetic.rs:146: the slot-ordered sum, from Rust's float Sum identity
            if (a != kAuxNone) sum = sum + ra.value(a & 63u, k);
            if (b != kAuxNone) sum = sum + rb.value(b & 63u, k);
            if (fresh[k]) post[size_t(fv) * G + k] = ch.v[k] + sum;
          }
        }
      }
    }
  }
  if (idle) return;
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

}  // namespace dev
}  // namespace ldpc
