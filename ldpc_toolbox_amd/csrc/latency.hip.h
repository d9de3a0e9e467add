// Small-batch (latency) path of the flooding min-sum decoder: the reference's own call pattern is ONE
// codeword per decode call (/root/reference/src/c_api/decoder.rs:50-67, src/simulation/ber.rs:462-466),
// where the batched kernels of kernels.hip.h -- lane = codeword -- would run with 1 lane in 64 useful
// and two launches per iteration.  Here the lanes go ACROSS THE ROWS AND VARIABLES OF ONE CODEWORD:
//
//   * one persistent launch per decode call; a codeword is owned by one XCD (32 CUs, one shared 4 MiB L2
//     in which the codeword's whole state and the graph tables stay resident: DVB-S2 n=64800 needs 3.7 MB),
//     up to 8 codewords of a call decode concurrently on the 8 XCDs, more take turns;
//   * check-node phase: a thread owns a check row (min1 / min2 / first argmin / sign mask in registers,
//     the same arithmetic as cn_minsum_kernel, so the results are bit-identical to the batch path);
//     variable-node phase: a thread owns a variable (slot-ordered sum from -0.0, arithmetic.rs:140-156);
//   * the rows are processed in the order of their first variable, and the graph tables and the messages
//     are stored in slices of 64 rows (64 variables), slot-major inside a slice (sliced ELLPACK): edge
//     (position p, slot j) has the id  slice_ptr[p / 64] + j * 64 + p % 64.  A wavefront owns a slice, so
//     its trip count is the slice's width (wave-uniform: no redundant loads, no divergence), every table /
//     message access of the check-node phase is one contiguous 256-byte segment, and in the structured
//     codes (DVB-S2's q-spaced checks, 5G's circulants) the gathers of both phases are mostly contiguous
//     too: the phases move little more than their algorithmic bytes through the L2;
//   * the phases are separated by a barrier over the XCD's workgroups only (one atomic counter).  Data that
//     crosses the barrier is written with plain (write-through) stores and read with nontemporal loads, which
//     bypass the per-CU L1: inside one XCD the shared L2 is then the point of coherence and no L1
//     invalidate / L2 write-back (1.7-6.5 us each, MI355X_MICROARCH.md) is needed.  Which XCD a workgroup
//     runs on is read from the hardware (XCC_ID), not assumed from the block index.
//
// Semantics per codeword are those of the batch path: pre-check on the raw input (iterations 0), stop at the
// first zero syndrome, -1 after max_iterations with the last hard decisions, max_iterations = 0 corner.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ldpc {
namespace dev {

struct LatencySync {
  uint32_t arrived[8];  // workgroups that reported in, per XCD
  uint32_t total;       // ... in all
  uint32_t pad0[7];
  uint32_t barrier[8];  // per-XCD barrier counters (monotonic)
  uint32_t unsat[8][2]; // per-XCD "some check is unsatisfied" flags, double-buffered by iteration parity
  uint32_t error;       // set when a bounded spin ran out (a workgroup never arrived): the results are invalid
};

struct LatencyTables {
  uint32_t n, m;
  uint32_t n_rslices, n_vslices;  // ceil(m / 64), ceil(n / 64)
  const uint32_t *rslice_ptr;     // [n_rslices+1] first edge id of a row slice (width = (next - this) / 64)
  const uint32_t *rdeg;           // [n_rslices*64] degree of the row at each position (0 beyond m)
  const uint32_t *col;            // [edge ids]     variable of the edge (0 where the row has no such slot)
  const uint32_t *vslice_ptr;     // [n_vslices+1]  first entry of a variable slice in vedge
  const uint32_t *vdeg;           // [n_vslices*64] degree of each variable (0 beyond n)
  const uint32_t *vedge;          // [entries]      edge id of the variable's k-th check, cols[v] order (0 where none)
  const int32_t *src_block;       // depuncture map or null
  uint32_t block_size;
};

struct LatencyState {  // 8 codeword slots (one per XCD) carved from one allocation: chan | post | msg | rawhard
  char *base;
  size_t slot_bytes, off_post, off_msg, off_rawhard;
};

// Experiment switches (tools/ab_variants.sh builds; the defaults are what ships):
//   LAT_LOAD_MODE  0 nontemporal loads, 1 agent-scope (sc1) loads, 2 plain loads (L1-cached: NOT coherent, timing only)
//   LAT_SYNC_SCOPE the scope of the barrier's add and of the flag stores.  Workgroup scope = no sc1 bit: the
//                  read-modify-write is still performed in the XCD's L2 (never in an L1) and the line stays
//                  there, where the polls -- always sc1 loads, which bypass the L1 -- find it: 1.35 us per
//                  barrier against 2.3 us with agent-scope adds, whose lines leave the L2 (measured,
//                  tools/latency_probe.py).  Valid because every participant of a barrier is on ONE XCD.
#ifndef LAT_LOAD_MODE
#define LAT_LOAD_MODE 0
#endif
#ifndef LAT_SYNC_SCOPE
#define LAT_SYNC_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
// every spin is bounded (about a second): a barrier that can never complete ends the kernel with
// LatencySync::error set instead of hanging the device
#define LAT_SPIN_LIMIT (1u << 20)

__device__ __forceinline__ float lat_load(const float *p) {
#if LAT_LOAD_MODE == 0
  return __builtin_nontemporal_load(p);  // bypasses the CU's L1: served by the XCD's L2
#elif LAT_LOAD_MODE == 1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}
__device__ __forceinline__ uint32_t lat_load(const uint8_t *p) {
#if LAT_LOAD_MODE == 2
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}

// polls always bypass the L1 (agent scope: an sc1 load, served by the L2 or beyond)
__device__ __forceinline__ uint32_t lat_atomic_load(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lat_atomic_store(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, LAT_SYNC_SCOPE);
}
// returns false when the spin ran out (and stays false: `*dead` short-cuts every later spin of this thread)
__device__ __forceinline__ void lat_spin_until(const uint32_t *p, uint32_t target, uint32_t *error, bool *dead) {
  if (*dead) return;
  uint32_t spins = 0;
  while (lat_atomic_load(p) < target) {
    if (++spins > LAT_SPIN_LIMIT) {
      __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *dead = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

struct LatEpoch {
  uint32_t passed = 0;  // barriers this workgroup has passed
  bool dead = false;    // a spin of this thread timed out
};

// Barrier over the `count` workgroups of this XCD; `*epoch` counts the barriers this workgroup has passed.
// raise / flag: when any thread of the workgroup passes raise != 0, *flag is set to 1 before the workgroup
// reports in (one store per workgroup), so every workgroup sees it after the barrier.
__device__ __forceinline__ void xcd_barrier(uint32_t *counter, uint32_t count, LatEpoch *epoch, uint32_t *error,
                                            uint32_t raise = 0, uint32_t *flag = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
  const int any = __syncthreads_or(static_cast<int>(raise));
  epoch->passed += 1;
  if (threadIdx.x == 0) {
    if (any && flag) {
      lat_atomic_store(flag, 1u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, LAT_SYNC_SCOPE);
    lat_spin_until(counter, epoch->passed * count, error, &epoch->dead);
  }
  __syncthreads();
}

// Check-node phase of one codeword.  A wavefront takes row slices w0, w0 + nwaves, ...; lane = row.  Writes
// the messages of this iteration (unless !WRITE) and returns whether any of the lane's rows has odd parity
// over the hard decisions of the previous posterior (FIRST: of the raw input; the messages then come from the
// channel LLRs).  U slots' loads are in flight together; the slot loop's bounds are wave-uniform.
template <bool FIRST, bool WRITE>
__device__ __forceinline__ uint32_t latency_cn_phase(const LatencyTables &g, const float *__restrict__ soft,
                                                     const uint8_t *__restrict__ rawhard, float *__restrict__ msg,
                                                     uint32_t w0, uint32_t nwaves, uint32_t lane) {
  constexpr uint32_t U = 8;
  const float inf = __builtin_huge_valf();
  uint32_t odd = 0;
  for (uint32_t sl = w0; sl < g.n_rslices; sl += nwaves) {
    const uint32_t e0 = g.rslice_ptr[sl], width = (g.rslice_ptr[sl + 1] - e0) >> 6;  // wave-uniform
    const uint32_t d = g.rdeg[sl * 64 + lane];
    const uint32_t base = e0 + lane;
    float min1 = inf, min2 = inf;
    uint32_t arg = 0, par = 0;
    uint64_t sgn = 0;
    for (uint32_t j0 = 0; j0 < width; j0 += U) {
      uint32_t vs[U];
      float ls[U], ms[U];
      uint32_t hs[U];
#pragma unroll
      for (uint32_t u = 0; u < U; u++)
        if (j0 + u < width) vs[u] = g.col[base + (j0 + u) * 64];
#pragma unroll
      for (uint32_t u = 0; u < U; u++) {
        if (j0 + u < width) {
          ls[u] = lat_load(soft + vs[u]);
          if (FIRST)
            hs[u] = lat_load(rawhard + vs[u]);
          else
            ms[u] = lat_load(msg + base + (j0 + u) * 64);
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < U; u++) {
        if (j0 + u < width) {
          const uint32_t j = j0 + u;
          const bool on = j < d;
          const float l = ls[u];
          const float x = FIRST ? l : (l - ms[u]);
          const float a = on ? fabsf(x) : inf;  // an absent slot never lowers a minimum
          par ^= on ? (FIRST ? hs[u] : (l <= 0.0f ? 1u : 0u)) : 0u;
          if (on && x < 0.0f) sgn |= uint64_t(1) << j;
          if (a < min1) {
            min2 = min1;
            min1 = a;
            arg = j;
          } else if (a < min2) {
            min2 = a;
          }
        }
      }
    }
    odd |= par;
    if (WRITE) {
      const uint32_t tot = __popcll(sgn) & 1u;
      for (uint32_t j = 0; j < width; j++) {
        const uint32_t neg = uint32_t(sgn >> j) & 1u;
        const float mag = (arg == j) ? min2 : min1;
        if (j < d) msg[base + j * 64] = (tot ^ neg) ? -mag : mag;
      }
    }
  }
  return odd;
}

// Variable-node phase: slot-ordered sum from -0.0, posterior = channel + sum (arithmetic.rs:140-156).  A
// wavefront takes variable slices; lane = variable.
__device__ __forceinline__ void latency_vn_phase(const LatencyTables &g, const float *__restrict__ chan,
                                                 const float *__restrict__ msg, float *__restrict__ post, uint32_t w0,
                                                 uint32_t nwaves, uint32_t lane) {
  constexpr uint32_t U = 8;
  for (uint32_t sl = w0; sl < g.n_vslices; sl += nwaves) {
    const uint32_t k0 = g.vslice_ptr[sl], width = (g.vslice_ptr[sl + 1] - k0) >> 6;  // wave-uniform
    const uint32_t v = sl * 64 + lane;
    const uint32_t d = g.vdeg[v];
    const uint32_t base = k0 + lane;
    const float c = v < g.n ? lat_load(chan + v) : 0.0f;
    float s = -0.0f;
    for (uint32_t j0 = 0; j0 < width; j0 += U) {
      uint32_t es[U];
      float ms[U];
#pragma unroll
      for (uint32_t u = 0; u < U; u++)
        if (j0 + u < width) es[u] = g.vedge[base + (j0 + u) * 64];
#pragma unroll
      for (uint32_t u = 0; u < U; u++)
        if (j0 + u < width) ms[u] = lat_load(msg + es[u]);
#pragma unroll
      for (uint32_t u = 0; u < U; u++)
        if (j0 + u < width && j0 + u < d) s = s + ms[u];
    }
    if (v < g.n) post[v] = c + s;
  }
}

template <typename SrcT>
__global__ __launch_bounds__(1024) void latency_minsum_kernel(LatencyTables g, LatencyState slots,
                                                              LatencySync *sync, const SrcT *__restrict__ llrs,
                                                              uint32_t input_len, uint32_t batch, uint32_t max_iterations,
                                                              uint8_t *__restrict__ bits, uint32_t out_len,
                                                              int32_t *__restrict__ iterations,
                                                              SrcT *__restrict__ posterior, uint32_t debug_skip) {
  // debug_skip (tools/latency_probe.py only; results are wrong when non-zero): bit 0 skips the check-node
  // work, bit 1 the variable-node work -- to time what is left (barriers are never skipped: a workgroup that
  // ran ahead would take another exit and strand the others)
  __shared__ uint32_t s_slot, s_count, s_rank, s_nx;
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 0) {
    s_slot = __hip_atomic_fetch_add(&sync->arrived[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&sync->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // census: every workgroup of the grid is resident and has reported (the grid is sized to fit)
    bool dead = false;
    lat_spin_until(&sync->total, gridDim.x, &sync->error, &dead);
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
  const uint32_t nwaves = nthreads >> 6, w0 = __builtin_amdgcn_readfirstlane(t0 >> 6), lane = threadIdx.x & 63u;
  LatEpoch epoch;
  uint32_t *const bar = &sync->barrier[xcc];
  const uint32_t n = g.n;

  // the XCDs that have workgroups share the codewords round-robin
  for (uint32_t cw = s_rank; cw < batch; cw += s_nx) {
    char *const slot = slots.base + size_t(s_rank) * slots.slot_bytes;
    float *__restrict__ chan = reinterpret_cast<float *>(slot);
    float *__restrict__ post = reinterpret_cast<float *>(slot + slots.off_post);
    float *__restrict__ msg = reinterpret_cast<float *>(slot + slots.off_msg);
    uint8_t *__restrict__ rawhard = reinterpret_cast<uint8_t *>(slot + slots.off_rawhard);
    const SrcT *src = llrs + size_t(cw) * input_len;
    uint32_t *const unsat = sync->unsat[xcc];

    // ingest: depuncture (puncturing.rs:83-101), quantise (`x as f32`), raw hard decisions for the pre-check
    for (uint32_t v = t0; v < n; v += nthreads) {
      SrcT raw;
      if (g.src_block) {
        const int32_t sb = g.src_block[v / g.block_size];
        raw = sb < 0 ? SrcT(0.0) : src[size_t(sb) * g.block_size + v % g.block_size];
      } else {
        raw = src[v];
      }
      chan[v] = static_cast<float>(raw);
      rawhard[v] = raw <= SrcT(0.0) ? 1 : 0;
    }
    if (t0 == 0) {
      lat_atomic_store(&unsat[0], 0u);
      lat_atomic_store(&unsat[1], 0u);
    }
    xcd_barrier(bar, count, &epoch, &sync->error);

    int32_t result = -1;  // iterations on success
    for (uint32_t it = 1; it <= max_iterations + 1; it++) {
      const bool first = it == 1, last = it == max_iterations + 1;
      // check nodes: messages of iteration `it` (not when `last`) and the parity of the previous posterior's
      // hard decisions over every row (the raw input's when `first`)
      uint32_t odd = 1;
      if (debug_skip & 1u) {
      } else if (first)
        odd = last ? latency_cn_phase<true, false>(g, chan, rawhard, msg, w0, nwaves, lane)
                   : latency_cn_phase<true, true>(g, chan, rawhard, msg, w0, nwaves, lane);
      else
        odd = last ? latency_cn_phase<false, false>(g, post, rawhard, msg, w0, nwaves, lane)
                   : latency_cn_phase<false, true>(g, post, rawhard, msg, w0, nwaves, lane);
      xcd_barrier(bar, count, &epoch, &sync->error, odd, &unsat[it & 1u]);
      const bool converged = lat_atomic_load(&unsat[it & 1u]) == 0;
      if (t0 == 0) lat_atomic_store(&unsat[(it + 1) & 1u], 0u);
      if (converged) {
        result = static_cast<int32_t>(it) - 1;  // flooding.rs:57-64 (0) / 69-79
        break;
      }
      if (last) break;
      if (!(debug_skip & 2u)) latency_vn_phase(g, chan, msg, post, w0, nwaves, lane);
      xcd_barrier(bar, count, &epoch, &sync->error);
    }

    // emit: converged at 0 -> the raw input's hard decisions and the (quantised) input; max_iterations = 0 and
    // not a codeword -> the reference's never-written output_llrs (all ones, 0.0); else hard(posterior)
    const bool zero_fill = result < 0 && max_iterations == 0;
    for (uint32_t v = t0; v < n; v += nthreads) {
      float val;
      uint8_t bit;
      if (result == 0) {
        val = lat_load(chan + v);
        bit = lat_load(rawhard + v);
      } else if (zero_fill) {
        val = 0.0f;
        bit = 1;
      } else {
        val = lat_load(post + v);
        bit = val <= 0.0f ? 1 : 0;
      }
      if (v < out_len) bits[size_t(cw) * out_len + v] = bit;
      if (posterior) posterior[size_t(cw) * n + v] = static_cast<SrcT>(val);
    }
    if (t0 == 0 && iterations) iterations[cw] = result;
    xcd_barrier(bar, count, &epoch, &sync->error);  // the slot's arrays are reused by this XCD's next codeword
  }
}

}  // namespace dev
}  // namespace ldpc
