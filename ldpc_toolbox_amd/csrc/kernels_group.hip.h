// A group's bookkeeping: init / latch, packed hard decisions and syndromes, ingest / emit (64 x 64 transposes through LDS),
// the syndrome operator, batch compaction.  Part of kernels.hip.h (include that).
#pragma once
namespace ldpc {
namespace dev {

// ---------------------------------------------------------------------------------------
// Bookkeeping kernels
// ---------------------------------------------------------------------------------------
#ifdef LDPC_GROUP_KERNELS_TU  // not a template: compiled in ONE translation unit (device_decoder.hip), launched through grp:: (device_decoder_internal.h)
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
#endif  // LDPC_GROUP_KERNELS_TU

// A codeword whose syndrome flag stayed clear is finished at `iteration`
// (flooding.rs:57-64, 69-79; horizontal_layered.rs:55-62, 66-78).
#ifdef LDPC_GROUP_KERNELS_TU  // not a template: compiled in ONE translation unit (device_decoder.hip), launched through grp:: (device_decoder_internal.h)
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
#endif  // LDPC_GROUP_KERNELS_TU

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
#ifdef LDPC_GROUP_KERNELS_TU  // not a template: compiled in ONE translation unit (device_decoder.hip), launched through grp:: (device_decoder_internal.h)
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
#endif  // LDPC_GROUP_KERNELS_TU

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
#ifdef LDPC_GROUP_KERNELS_TU  // not a template: compiled in ONE translation unit (device_decoder.hip), launched through grp:: (device_decoder_internal.h)
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
#endif  // LDPC_GROUP_KERNELS_TU

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
#ifdef LDPC_GROUP_KERNELS_TU  // not a template: compiled in ONE translation unit (device_decoder.hip), launched through grp:: (device_decoder_internal.h)
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
#endif  // LDPC_GROUP_KERNELS_TU

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

#ifdef LDPC_GROUP_KERNELS_TU  // not a template: compiled in ONE translation unit (device_decoder.hip), launched through grp:: (device_decoder_internal.h)
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
#endif  // LDPC_GROUP_KERNELS_TU

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

#ifdef LDPC_GROUP_KERNELS_TU  // straggler pooling of the batch entries (DeviceDecoder::decode_device_pooled): three small kernels
// Iteration counts of a chunk -> the call's statistics and the indices of the chunk's stragglers.
//   stats[0] += frames that converged, stats[1] += frames that failed with the FULL budget, stats[2] = stragglers so far
//   (= next free entry of idx), sum_its += iterations of the converged frames.  reduced: the chunk ran a reduced budget,
//   so a -1 means "not yet" and the frame's index (first + i) is appended to idx.
__global__ __launch_bounds__(256) void pool_select_kernel(const int32_t *__restrict__ its, uint32_t nf, uint32_t first, int reduced,
                                                          uint32_t *__restrict__ idx, uint32_t *__restrict__ stats,
                                                          unsigned long long *__restrict__ sum_its) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t it = i < nf ? its[i] : 0;
  const bool ok = i < nf && it >= 0, miss = i < nf && it < 0;
  const uint64_t ok_mask = __builtin_amdgcn_ballot_w64(ok), miss_mask = __builtin_amdgcn_ballot_w64(miss);
  const uint32_t lane = threadIdx.x & 63u;
  // one atomic per wavefront and counter
  unsigned long long s = ok ? static_cast<unsigned long long>(it) : 0ull;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  uint32_t base = 0;
  if (lane == 0) {
    if (ok_mask) {
      atomicAdd(&stats[0], static_cast<uint32_t>(__builtin_popcountll(ok_mask)));
      atomicAdd(sum_its, s);
    }
    if (miss_mask) base = atomicAdd(&stats[reduced ? 2 : 1], static_cast<uint32_t>(__builtin_popcountll(miss_mask)));
  }
  base = __shfl(base, 0, 64);
  if (miss && reduced) idx[base + __builtin_popcountll(miss_mask & ((1ull << lane) - 1ull))] = first + i;
}
// rows by index, a wavefront per row: dst[i] = src[idx[i]]  (rows of row_words 32-bit words)
__global__ __launch_bounds__(256) void pool_gather_kernel(const uint32_t *__restrict__ idx, uint32_t count,
                                                          const uint32_t *__restrict__ src, size_t row_words,
                                                          uint32_t *__restrict__ dst) {
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (i >= count) return;
  const uint32_t *s = src + size_t(idx[i]) * row_words;
  uint32_t *d = dst + size_t(i) * row_words;
  for (size_t w = lane; w < row_words; w += 64) d[w] = s[w];
}
// results of pooled frame i -> frame idx[i]: hard decisions (bytes), iteration count, posterior row (may be null)
__global__ __launch_bounds__(256) void pool_scatter_kernel(const uint32_t *__restrict__ idx, uint32_t count,
                                                           const uint8_t *__restrict__ pbits, uint32_t out_len,
                                                           const int32_t *__restrict__ pits, const uint32_t *__restrict__ ppost,
                                                           size_t post_words, uint8_t *__restrict__ bits,
                                                           int32_t *__restrict__ its, uint32_t *__restrict__ post) {
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (i >= count) return;
  const size_t f = idx[i];
  for (uint32_t b = lane; b < out_len; b += 64) bits[f * out_len + b] = pbits[size_t(i) * out_len + b];
  if (lane == 0 && its != nullptr) its[f] = pits[i];
  if (post != nullptr)
    for (size_t w = lane; w < post_words; w += 64) post[f * post_words + w] = ppost[size_t(i) * post_words + w];
}
#endif  // LDPC_GROUP_KERNELS_TU

}  // namespace dev
}  // namespace ldpc
