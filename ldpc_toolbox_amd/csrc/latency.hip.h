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
//   * the rows are processed in the order of their first variable, the variables are renumbered in the order
//     of their first appearance in a slot-major scan of those rows, and the graph tables and the messages
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
  uint32_t barrier_unused[8];
  // per-XCD barrier words (monotonic), 8 shards each on a 128-byte line of its own: the arrivals of an XCD's 32
  // workgroups on ONE word serialise in the L2 (1.25 us per barrier); workgroup slot s reports to shard s % 8
  // and eight lanes of the waiting wavefront poll one shard each.  A word carries the arrivals (bits 0-31)
  // and, for the barriers that also vote ("does any workgroup raise its hand?"), two 16-bit vote counters
  // (bits 32-47: even barriers, 48-63: odd ones -- a workgroup can be at most one barrier ahead of another),
  // so the vote costs no store, no wait and no load of its own.
  uint64_t barrier[8][8][16];
  uint32_t error;       // set when a bounded spin ran out (a workgroup never arrived): the results are invalid
};

struct LatencyTables {
  uint32_t n, m;
  uint32_t n_rslices, n_vslices;  // ceil(m / 64), ceil(n / 64)
  const uint32_t *rslice_ptr;     // [n_rslices+1] first edge id of a row slice (width = (next - this) / 64)
  const uint32_t *rdeg;           // [n_rslices*64] degree of the row at each position (0 beyond m)
  const uint32_t *col;            // [edge ids + 512] the edge's variable, as its perm index (0 where the row has no such slot)
  const uint32_t *vslice_ptr;     // [n_vslices+1]  first entry of a variable slice in vedge
  const uint32_t *vdeg;           // [n_vslices*64] degree of the variable at each perm index (0 beyond n)
  const uint32_t *vedge;          // [entries + 512] edge id of the variable's k-th check, cols[v] order (0 where none)
  const uint32_t *perm;           // [n] variable -> its index in the per-codeword arrays (first-appearance order)
  const uint32_t *inv;            // [n] the inverse
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
// every spin is bounded (some 50 ms): a barrier that cannot complete -- workgroups of the grid that do not
// become resident because something else holds the CUs -- ends the kernel with LatencySync::error set instead
// of hanging the device, and the host redoes the call with the batched kernels
#define LAT_SPIN_LIMIT (1u << 16)

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

// Buffer addressing for the per-codeword arrays: a wave-uniform descriptor (SGPRs) + a 32-bit byte offset per
// lane, so a gather costs one VGPR and no 64-bit address arithmetic.  Loads carry the nontemporal bit (aux 2)
// like lat_load; stores are plain (write-through to the L2).
#ifdef LAT_NO_BUFFER  // bisecting aid: the same accessors over plain global pointers
struct LatBuf {
  char *p;
};
__device__ __forceinline__ LatBuf lat_buf(const void *p, uint32_t) { return LatBuf{const_cast<char *>(static_cast<const char *>(p))}; }
__device__ __forceinline__ float lat_bload(const LatBuf &b, uint32_t byte_off, uint32_t soff = 0) {
  return __builtin_nontemporal_load(reinterpret_cast<const float *>(b.p + byte_off + soff));
}
__device__ __forceinline__ uint32_t lat_bload_u8(const LatBuf &b, uint32_t byte_off) {
  return __builtin_nontemporal_load(reinterpret_cast<const uint8_t *>(b.p + byte_off));
}
__device__ __forceinline__ void lat_bstore(const LatBuf &b, uint32_t byte_off, uint32_t soff, float v) {
  *reinterpret_cast<float *>(b.p + byte_off + soff) = v;
}
#else
struct LatBuf {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ LatBuf lat_buf(const void *p, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  // (readfirstlane returns int: widen through uint32_t, or a low word with bit 31 set sign-extends into the high one)
  const uint64_t u = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(a >> 32)))) << 32) |
                     uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(a))));
  return LatBuf{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(u), 0, static_cast<int>(bytes), 0x00020000)};
}
__device__ __forceinline__ float lat_bload(const LatBuf &b, uint32_t byte_off, uint32_t soff = 0) {
#if LAT_LOAD_MODE == 2
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, byte_off, soff, 0));
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, byte_off, soff, 2));
#endif
}
__device__ __forceinline__ uint32_t lat_bload_u8(const LatBuf &b, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b8(b.r, byte_off, 0, 2);
}
__device__ __forceinline__ void lat_bstore(const LatBuf &b, uint32_t byte_off, uint32_t soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), b.r, byte_off, soff, 0);
}
#endif

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
      __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      *dead = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

struct LatEpoch {
  uint32_t passed = 0;  // barriers this workgroup has passed
  uint32_t seen = 0;    // (lanes 0-7 of the first wavefront) the two vote counters of "my" shard as last read
  bool dead = false;    // a spin of this thread timed out
};

// Barrier over the `count` workgroups of this XCD (slot = this workgroup's index among them) that also returns
// whether ANY thread of ANY of them passed raise != 0.
__device__ __forceinline__ bool xcd_barrier(uint64_t (*shards)[16], uint32_t count, uint32_t slot, LatEpoch *epoch,
                                            uint32_t *error, uint32_t raise = 0) {
  __shared__ uint32_t s_vote;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
  const int any = __syncthreads_or(static_cast<int>(raise));
  const uint32_t parity = epoch->passed & 1u;
  epoch->passed += 1;
  if (threadIdx.x < 64) {  // the workgroup's first wavefront
    const uint32_t lane = threadIdx.x;
    if (lane == 0)
      __hip_atomic_fetch_add(&shards[slot & 7u][0], uint64_t(1) | (any ? (uint64_t(1) << (32 + 16 * parity)) : 0),
                             __ATOMIC_RELAXED, LAT_SYNC_SCOPE);
    // lane k < 8 watches shard k: complete when it has seen passed * (workgroups reporting to it) arrivals
    const uint32_t mine = lane < 8 ? (count + 7u - lane) / 8u : 0u;
    const uint32_t target = epoch->passed * mine;
    uint64_t w = 0;
    uint32_t spins = 0;
    while (!epoch->dead) {
      if (mine != 0) w = __hip_atomic_load(&shards[lane & 7u][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool ok = mine == 0 || uint32_t(w) >= target;
      if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
      if (++spins > LAT_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        epoch->dead = true;
      }
    }
    // votes of this barrier: my shard's counter of this parity moved since I last looked
    const uint32_t votes = uint32_t(w >> 32);
    const uint32_t field = (votes >> (16 * parity)) & 0xFFFFu, before = (epoch->seen >> (16 * parity)) & 0xFFFFu;
    const bool moved = mine != 0 && field != before;
    // only this parity's counter is final now: the other one may already hold votes of the NEXT barrier from
    // workgroups that are ahead, and must be compared with its value at the end of ITS last barrier
    epoch->seen = (epoch->seen & ~(0xFFFFu << (16 * parity))) | (field << (16 * parity));
    const uint64_t raised = __builtin_amdgcn_ballot_w64(moved);
    if (lane == 0) s_vote = raised != 0 ? 1u : 0u;
  }
  __syncthreads();
  return s_vote != 0;
}

__device__ __forceinline__ uint32_t lat_uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// A wavefront works on the same slices in every iteration, so the graph indices of its first row slice and
// first two variable slices (8 slots each; further slots and further slices are re-read from the tables) are
// loaded once per call and kept in registers, as byte offsets: an iteration's phases then consist of ONE
// round trip of data loads each (soft values + messages; messages), the computation, and the stores.
struct LatRowCache {
  uint32_t col4[8], deg, e0, width;  // col4 = 4 * variable; e0, width wave-uniform
};
struct LatVarCache {
  uint32_t edge4[8], deg, k0, width;  // edge4 = 4 * edge id
  float chan;
};

struct LatCnAcc {
  float min1, min2;
  uint32_t arg, par;
  uint64_t sgn;
};

// the per-codeword arrays as buffers
struct LatArrays {
  LatBuf soft, msg, rawhard;  // soft: chan (first iteration) or post
};

// U slots of a row: loads (all in flight together), then the running min1 / min2 / first argmin / signs / parity.
// The loads are UNCONDITIONAL: the tables and the message array are padded by 8 * 64 entries, so a chunk that
// runs past the slice's width reads valid (next slice's) entries, and the lane's degree masks the results.
// vs4: byte offsets of the variables; base4: byte offset of the lane's slot-0 message; j0: first slot.
template <uint32_t U, bool FIRST>
__device__ __forceinline__ void latency_cn_chunk(LatCnAcc &acc, const uint32_t (&vs4)[U], const LatArrays &A, uint32_t base4,
                                                 uint32_t j0, uint32_t d) {
  const float inf = __builtin_huge_valf();
  float ls[U], ms[U];
  uint32_t hs[U];
  const uint32_t soff = lat_uniform(j0 * 256);
#pragma unroll
  for (uint32_t u = 0; u < U; u++) {
    ls[u] = lat_bload(A.soft, vs4[u]);
    if (FIRST)
      hs[u] = lat_bload_u8(A.rawhard, vs4[u] >> 2);
    else
      ms[u] = lat_bload(A.msg, base4 + u * 256, soff);
  }
#pragma unroll
  for (uint32_t u = 0; u < U; u++) {
    const uint32_t j = j0 + u;
    const bool on = j < d;
    const float l = ls[u];
    const float x = FIRST ? l : (l - ms[u]);
    const float a = on ? fabsf(x) : inf;  // an absent slot never lowers a minimum
    acc.par ^= on ? (FIRST ? hs[u] : (l <= 0.0f ? 1u : 0u)) : 0u;
    if (on && x < 0.0f) acc.sgn |= uint64_t(1) << j;
    if (a < acc.min1) {
      acc.min2 = acc.min1;
      acc.min1 = a;
      acc.arg = j;
    } else if (a < acc.min2) {
      acc.min2 = a;
    }
  }
}

// One row slice of the check-node phase.  cache: the slice's first 8 slots' variables in registers, or null.
template <uint32_t U, bool FIRST, bool WRITE>
__device__ __forceinline__ uint32_t latency_cn_slice(const LatencyTables &g, const LatArrays &A, uint32_t e0, uint32_t width,
                                                     uint32_t d, uint32_t lane, const LatRowCache *cache) {
  const uint32_t base = e0 + lane, base4 = base * 4;
  LatCnAcc acc{__builtin_huge_valf(), __builtin_huge_valf(), 0, 0, 0};
  uint32_t j0 = 0;
  if (cache) {
#pragma unroll
    for (uint32_t c = 0; c < 8 / U; c++) {
      if (c * U < width) {  // wave-uniform
        uint32_t vs4[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) vs4[u] = cache->col4[c * U + u];
        latency_cn_chunk<U, FIRST>(acc, vs4, A, base4, c * U, d);
      }
    }
    j0 = 8;
  }
  for (; j0 < width; j0 += U) {
    uint32_t vs4[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) vs4[u] = g.col[base + (j0 + u) * 64] * 4;
    latency_cn_chunk<U, FIRST>(acc, vs4, A, base4, j0, d);
  }
  if (WRITE) {
    const uint32_t tot = __popcll(acc.sgn) & 1u;
    for (uint32_t j = 0; j < width; j++) {  // wave-uniform trip count; j * 256 rides in the scalar offset
      const uint32_t neg = uint32_t(acc.sgn >> j) & 1u;
      const float mag = (acc.arg == j) ? acc.min2 : acc.min1;
      if (j < d) lat_bstore(A.msg, base4, lat_uniform(j * 256), (tot ^ neg) ? -mag : mag);
    }
  }
  return acc.par;
}

template <bool FIRST, bool WRITE>
__device__ __forceinline__ uint32_t latency_cn_any(const LatencyTables &g, const LatArrays &A, uint32_t e0, uint32_t width,
                                                   uint32_t d, uint32_t lane, const LatRowCache *cache) {
  if (width <= 4) return latency_cn_slice<4, FIRST, WRITE>(g, A, e0, width, d, lane, cache);
  return latency_cn_slice<8, FIRST, WRITE>(g, A, e0, width, d, lane, cache);
}

// Check-node phase of one codeword.  A wavefront takes row slices w0, w0 + nwaves, ...; lane = row.  Writes
// the messages of this iteration (unless !WRITE) and returns whether any of the lane's rows has odd parity
// over the hard decisions of the previous posterior (FIRST: of the raw input; the messages then come from the
// channel LLRs).  The chunk size follows the slice's width (wave-uniform).
template <bool FIRST, bool WRITE>
__device__ __forceinline__ uint32_t latency_cn_phase(const LatencyTables &g, const LatArrays &A, uint32_t w0, uint32_t nwaves,
                                                     uint32_t lane, const LatRowCache &rc) {
  uint32_t odd = 0;
#ifdef LAT_NO_CACHE
  if (w0 < g.n_rslices) odd |= latency_cn_any<FIRST, WRITE>(g, A, rc.e0, rc.width, rc.deg, lane, nullptr);
#else
  if (w0 < g.n_rslices) odd |= latency_cn_any<FIRST, WRITE>(g, A, rc.e0, rc.width, rc.deg, lane, &rc);
#endif
  for (uint32_t sl = w0 + nwaves; sl < g.n_rslices; sl += nwaves) {
    const uint32_t e0 = lat_uniform(g.rslice_ptr[sl]), width = lat_uniform((g.rslice_ptr[sl + 1] - e0) >> 6);
    const uint32_t d = g.rdeg[sl * 64 + lane];
    odd |= latency_cn_any<FIRST, WRITE>(g, A, e0, width, d, lane, nullptr);
  }
  return odd;
}

template <uint32_t U>
__device__ __forceinline__ void latency_vn_chunk(float &s, const uint32_t (&es4)[U], const LatBuf &msg, uint32_t j0,
                                                 uint32_t d) {
  float ms[U];
#pragma unroll
  for (uint32_t u = 0; u < U; u++) ms[u] = lat_bload(msg, es4[u]);
#pragma unroll
  for (uint32_t u = 0; u < U; u++)
    if (j0 + u < d) s = s + ms[u];
}

// slot-ordered sum from -0.0 of one variable slice's messages (arithmetic.rs:140-156)
template <uint32_t U>
__device__ __forceinline__ float latency_vn_slice(const LatencyTables &g, const LatBuf &msg, uint32_t k0, uint32_t width,
                                                  uint32_t d, uint32_t lane, const LatVarCache *cache) {
  const uint32_t base = k0 + lane;
  float s = -0.0f;
  uint32_t j0 = 0;
  if (cache) {
#pragma unroll
    for (uint32_t c = 0; c < 8 / U; c++) {
      if (c * U < width) {  // wave-uniform
        uint32_t es4[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) es4[u] = cache->edge4[c * U + u];
        latency_vn_chunk<U>(s, es4, msg, c * U, d);
      }
    }
    j0 = 8;
  }
  for (; j0 < width; j0 += U) {
    uint32_t es4[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) es4[u] = g.vedge[base + (j0 + u) * 64] * 4;
    latency_vn_chunk<U>(s, es4, msg, j0, d);
  }
  return s;
}

__device__ __forceinline__ float latency_vn_any(const LatencyTables &g, const LatBuf &msg, uint32_t k0, uint32_t width,
                                                uint32_t d, uint32_t lane, const LatVarCache *cache) {
  if (width <= 2) return latency_vn_slice<2>(g, msg, k0, width, d, lane, cache);
  if (width <= 4) return latency_vn_slice<4>(g, msg, k0, width, d, lane, cache);
  return latency_vn_slice<8>(g, msg, k0, width, d, lane, cache);
}

// Variable-node phase: posterior = channel + sum.  A wavefront takes variable slices; lane = variable.  The
// two register-cached slices' first 8 messages are all requested before either sum starts.
__device__ __forceinline__ void latency_vn_phase(const LatencyTables &g, const float *__restrict__ chan, const LatBuf &msg,
                                                 float *__restrict__ post, uint32_t w0, uint32_t nwaves, uint32_t lane,
                                                 const LatVarCache (&vc)[2]) {
#ifdef LAT_NO_CACHE
#pragma unroll
  for (uint32_t i = 0; i < 2; i++) {
    const uint32_t sl = w0 + i * nwaves;
    if (sl < g.n_vslices) {
      const uint32_t v = sl * 64 + lane;
      const float s = latency_vn_any(g, msg, vc[i].k0, vc[i].width, vc[i].deg, lane, nullptr);
      if (v < g.n) post[v] = vc[i].chan + s;
    }
  }
#else
  float ms[2][8];
#pragma unroll
  for (uint32_t i = 0; i < 2; i++) {
    if (w0 + i * nwaves < g.n_vslices) {  // wave-uniform
#pragma unroll
      for (uint32_t u = 0; u < 8; u++)
        if (u < vc[i].width) ms[i][u] = lat_bload(msg, vc[i].edge4[u]);  // wave-uniform guard
    }
  }
#pragma unroll
  for (uint32_t i = 0; i < 2; i++) {
    const uint32_t sl = w0 + i * nwaves;
    if (sl < g.n_vslices) {
      float s = -0.0f;
#pragma unroll
      for (uint32_t u = 0; u < 8; u++)
        if (u < vc[i].width && u < vc[i].deg) s = s + ms[i][u];
      if (vc[i].width > 8) {  // slots beyond the cached ones
        const uint32_t base = vc[i].k0 + lane;
        for (uint32_t j0 = 8; j0 < vc[i].width; j0 += 8) {
          uint32_t es4[8];
#pragma unroll
          for (uint32_t u = 0; u < 8; u++) es4[u] = g.vedge[base + (j0 + u) * 64] * 4;
          latency_vn_chunk<8>(s, es4, msg, j0, vc[i].deg);
        }
      }
      const uint32_t v = sl * 64 + lane;
      if (v < g.n) post[v] = vc[i].chan + s;
    }
  }
#endif
  for (uint32_t sl = w0 + 2 * nwaves; sl < g.n_vslices; sl += nwaves) {
    const uint32_t k0 = lat_uniform(g.vslice_ptr[sl]), width = lat_uniform((g.vslice_ptr[sl + 1] - k0) >> 6);
    const uint32_t v = sl * 64 + lane;
    const uint32_t d = g.vdeg[v];
    const float c = lat_load(chan + min(v, g.n - 1));
    const float s = latency_vn_any(g, msg, k0, width, d, lane, nullptr);
    if (v < g.n) post[v] = c + s;
  }
}

template <typename SrcT>
__global__ __launch_bounds__(1024) void latency_minsum_kernel(LatencyTables g, LatencyState slots,
                                                              LatencySync *sync, const SrcT *__restrict__ llrs,
                                                              uint32_t input_len, uint32_t batch, uint32_t max_iterations,
                                                              uint8_t *__restrict__ bits, uint32_t out_len,
                                                              int32_t *__restrict__ iterations,
                                                              SrcT *__restrict__ posterior, uint32_t *error_word LDPC_DBG_PARAM(debug_skip)) {
#ifndef LDPC_EXPERIMENTS
  constexpr uint32_t debug_skip = 0;
#endif
  // debug_skip (tools/latency_probe.py only; results are wrong when non-zero): bit 0 skips the check-node
  // work, bit 1 the variable-node work, bit 2 the check-node phase's message stores, bit 3 redirects the
  // posterior stores to a scratch row -- to time what is left (barriers are never skipped: a workgroup that
  // ran ahead would take another exit and strand the others)
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
  // this wavefront's first row slice and first two variable slices: indices into registers, once per call
  LatRowCache rc{};
  LatVarCache vc[2]{};
  if (w0 < g.n_rslices) {
    rc.e0 = lat_uniform(g.rslice_ptr[w0]);
    rc.width = lat_uniform((g.rslice_ptr[w0 + 1] - rc.e0) >> 6);
    rc.deg = g.rdeg[w0 * 64 + lane];
#pragma unroll
    for (uint32_t u = 0; u < 8; u++) rc.col4[u] = g.col[rc.e0 + lane + u * 64] * 4;
  }
#pragma unroll
  for (uint32_t i = 0; i < 2; i++) {
    const uint32_t sl = w0 + i * nwaves;
    if (sl < g.n_vslices) {
      vc[i].k0 = lat_uniform(g.vslice_ptr[sl]);
      vc[i].width = lat_uniform((g.vslice_ptr[sl + 1] - vc[i].k0) >> 6);
      vc[i].deg = g.vdeg[sl * 64 + lane];
#pragma unroll
      for (uint32_t u = 0; u < 8; u++) vc[i].edge4[u] = g.vedge[vc[i].k0 + lane + u * 64] * 4;
    }
  }

  // the XCDs that have workgroups share the codewords round-robin
  for (uint32_t cw = s_rank; cw < batch; cw += s_nx) {
    char *const slot = slots.base + size_t(s_rank) * slots.slot_bytes;
    float *__restrict__ chan = reinterpret_cast<float *>(slot);
    float *__restrict__ post = reinterpret_cast<float *>(slot + slots.off_post);
    float *__restrict__ msg = reinterpret_cast<float *>(slot + slots.off_msg);
    uint8_t *__restrict__ rawhard = reinterpret_cast<uint8_t *>(slot + slots.off_rawhard);
    const SrcT *src = llrs + size_t(cw) * input_len;
    const uint32_t soft_bytes = static_cast<uint32_t>(slots.off_post), msg_bytes = static_cast<uint32_t>(slots.off_rawhard - slots.off_msg);
    const LatBuf b_msg = lat_buf(msg, msg_bytes);
    const LatArrays a_first{lat_buf(chan, soft_bytes), b_msg, lat_buf(rawhard, static_cast<uint32_t>(slots.slot_bytes - slots.off_rawhard))};
    const LatArrays a_iter{lat_buf(post, soft_bytes), b_msg, a_first.rawhard};

    // ingest: depuncture (puncturing.rs:83-101), quantise (`x as f32`), raw hard decisions for the pre-check
    // (in source order: `llrs` may be the caller's pinned host buffer, read over the bus -- coalesced reads there,
    // the scatter lands in device memory)
    for (uint32_t v = t0; v < n; v += nthreads) {
      const uint32_t t = g.perm[v];
      SrcT raw;
      if (g.src_block) {
        const int32_t sb = g.src_block[v / g.block_size];
        raw = sb < 0 ? SrcT(0.0) : src[size_t(sb) * g.block_size + v % g.block_size];
      } else {
        raw = src[v];
      }
      chan[t] = static_cast<float>(raw);
      rawhard[t] = raw <= SrcT(0.0) ? 1 : 0;
    }
    xcd_barrier(bar, count, my_slot, &epoch, error_word);
#pragma unroll
    for (uint32_t i = 0; i < 2; i++) {
      const uint32_t sl = w0 + i * nwaves;
      if (sl < g.n_vslices) vc[i].chan = lat_load(chan + min(sl * 64 + lane, n - 1));
    }

    int32_t result = -1;  // iterations on success
    for (uint32_t it = 1; it <= max_iterations + 1; it++) {
      const bool first = it == 1, last = it == max_iterations + 1;
      // check nodes: messages of iteration `it` (not when `last`) and the parity of the previous posterior's
      // hard decisions over every row (the raw input's when `first`)
      uint32_t odd = 1;
      if (debug_skip & 1u) {
      } else if (first)
        odd = (last || (debug_skip & 4u)) ? latency_cn_phase<true, false>(g, a_first, w0, nwaves, lane, rc)
                   : latency_cn_phase<true, true>(g, a_first, w0, nwaves, lane, rc);
      else
        odd = (last || (debug_skip & 4u)) ? latency_cn_phase<false, false>(g, a_iter, w0, nwaves, lane, rc)
                   : latency_cn_phase<false, true>(g, a_iter, w0, nwaves, lane, rc);
      const bool converged = !xcd_barrier(bar, count, my_slot, &epoch, error_word, odd);  // no row anywhere is odd
      if (converged) {
        result = static_cast<int32_t>(it) - 1;  // flooding.rs:57-64 (0) / 69-79
        break;
      }
      if (last) break;
      if (!(debug_skip & 2u)) latency_vn_phase(g, chan, b_msg, (debug_skip & 8u) ? post + g.n + 64 : post, w0, nwaves, lane, vc);
      xcd_barrier(bar, count, my_slot, &epoch, error_word);
    }

    // emit: converged at 0 -> the raw input's hard decisions and the (quantised) input; max_iterations = 0 and
    // not a codeword -> the reference's never-written output_llrs (all ones, 0.0); else hard(posterior)
    const bool zero_fill = result < 0 && max_iterations == 0;
    for (uint32_t v = t0; v < n; v += nthreads) {
      const uint32_t t = g.perm[v];
      float val;
      uint8_t bit;
      if (result == 0) {
        val = lat_load(chan + t);
        bit = lat_load(rawhard + t);
      } else if (zero_fill) {
        val = 0.0f;
        bit = 1;
      } else {
        val = lat_load(post + t);
        bit = val <= 0.0f ? 1 : 0;
      }
      if (v < out_len) bits[size_t(cw) * out_len + v] = bit;
      if (posterior) posterior[size_t(cw) * n + v] = static_cast<SrcT>(val);
    }
    if (t0 == 0 && iterations) iterations[cw] = result;
    xcd_barrier(bar, count, my_slot, &epoch, error_word);  // the slot's arrays are reused by this XCD's next codeword
  }
}

}  // namespace dev
}  // namespace ldpc
