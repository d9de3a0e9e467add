// Internal to the library's translation units (device_decoder.hip, run_group_f32.hip, run_group_f64.hip,
// run_group_i8.hip, latency_paths.hip): the per-handle state behind DeviceDecoder's opaque members, launch geometry,
// the per-call knobs and the host's view of a group's progress word.  Round 6 cut the one 3 200-line translation unit
// (6 minutes of compile time) by what instantiates kernels: the f32 and f64 float rules, the 8-bit rules and the small-batch
// kernels each compile on their own, in parallel; every kernel is still instantiated in exactly one of them.
#pragma once
#include "device_decoder.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include "kernels.hip.h"
#include "slice_tasks.h"
#include "latency.hip.h"
#include "latency_edge.hip.h"

namespace ldpc {

inline uint32_t env_u32(const char *name, uint32_t dflt) {
  const char *s = std::getenv(name);
  if (!s || !*s) return dflt;
  return static_cast<uint32_t>(std::strtoul(s, nullptr, 10));
}

inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

struct DeviceDecoder::Workspace {
  size_t G = 0;  // codewords per group this workspace is sized for
  size_t elem = 4;
  void *slab = nullptr;  // one allocation; the arrays below are carved from it
  bool borrowed = false;  // the slab is a part of the decoder's joint allocation for both lanes (ensure_lanes)
  size_t slab_bytes = 0;
  void *chan = nullptr, *post = nullptr, *msg = nullptr, *msg2 = nullptr;
  void *rec[2] = {nullptr, nullptr};  // row records, double-buffered (instead of msg2)
  bool records = false;
  uint64_t *rawbits = nullptr, *hardbits = nullptr;
  // compaction: perm = the movers' slots, slot_tmp = the holes they fill, fill_cw = codeword landing in a slot
  uint32_t *perm = nullptr, *slot_cw = nullptr, *slot_tmp = nullptr, *fill_cw = nullptr, *n_slots = nullptr;
  dev::CompactPlan *plan = nullptr;
  uint32_t *done = nullptr, *unsat0 = nullptr, *unsat1 = nullptr, *n_active = nullptr, *scratch_flags = nullptr,
           *slice_state = nullptr, *it0 = nullptr, *holes = nullptr;
  dev::StreamPlan *stream_plan = nullptr;
  int32_t *iters = nullptr;
  // progress word (pinned host memory, mapped into the device): kernels.hip.h, State::publish
  uint64_t *h_flag = nullptr, *d_flag = nullptr;
  uint32_t epoch = 0;
  // device-side input staging of decode_host (one group's rows as the caller laid them out)
  void *in = nullptr;
  size_t in_bytes = 0;
  // decode_host: recorded right after the ingest kernel of the group being enqueued (the lane's input buffer is free
  // again), and counted, so that the staging thread knows the record has been made
  hipEvent_t after_ingest = nullptr;
  std::atomic<uint32_t> *ingest_seq = nullptr;
  // check rows too long for the LDS-staged kernels' columns (more than 160 KB per 64 threads): per-wavefront columns in
  // HBM, allocated at the first call that needs them (kernels.hip.h, cn_staged_kernel SCRATCH)
  void *row_scratch = nullptr;
  size_t row_scratch_bytes = 0;

  void release() {
    if (slab && !borrowed) (void)hipFree(slab);
    if (in) (void)hipFree(in);
    if (row_scratch) (void)hipFree(row_scratch);
    if (h_flag) (void)hipHostFree(h_flag);
    *this = Workspace();
  }
};

// pinned staging of the host-pointer entry (decode_host, further down)
struct DeviceDecoder::HostPipe {
  static constexpr size_t kChunk = size_t(32) << 20;
  static constexpr int kSlots = 4;
  static constexpr int kOutRing = 4;  // group-sized device output buffers (two per execution lane)
  // pinned chunks, allocated at first use and only as large as the calls need (a reference-style scalar call
  // pins a few hundred KB, not 8 x 32 MiB)
  char *in_slot[kSlots] = {}, *out_slot[kSlots] = {};
  size_t in_cap[kSlots] = {}, out_cap[kSlots] = {};
  hipEvent_t in_done[kSlots] = {}, out_done[kSlots] = {};
  int next_in = 0;
  hipStream_t h2d = nullptr, d2h = nullptr;
  hipEvent_t in_ready[2] = {}, ingested[2] = {};
  std::vector<hipEvent_t> group_done;
  uint8_t *d_bits[kOutRing] = {};
  int32_t *d_iters[kOutRing] = {};
  void *d_post[kOutRing] = {};
  size_t bits_cap[kOutRing] = {}, iters_cap[kOutRing] = {}, post_cap[kOutRing] = {};
  unsigned copy_threads = 1;

  // a pinned chunk of at least `need` bytes (<= kChunk) in *slot
  static int pinned(char **slot, size_t *cap, size_t need) {
    if (*cap >= need) return 0;
    if (*slot) (void)hipHostFree(*slot);
    *slot = nullptr;
    *cap = 0;
    const size_t bytes = std::min(kChunk, (need + (size_t(64) << 10) - 1) >> 16 << 16);
    if (hipHostMalloc(reinterpret_cast<void **>(slot), bytes, hipHostMallocDefault) != hipSuccess) return -2;
    *cap = bytes;
    return 0;
  }
  void release() {
    for (int i = 0; i < kSlots; i++) {
      if (in_slot[i]) (void)hipHostFree(in_slot[i]);
      if (out_slot[i]) (void)hipHostFree(out_slot[i]);
      if (in_done[i]) (void)hipEventDestroy(in_done[i]);
      if (out_done[i]) (void)hipEventDestroy(out_done[i]);
    }
    for (int l = 0; l < 2; l++) {
      if (in_ready[l]) (void)hipEventDestroy(in_ready[l]);
      if (ingested[l]) (void)hipEventDestroy(ingested[l]);
    }
    for (auto e : group_done) (void)hipEventDestroy(e);
    if (h2d) (void)hipStreamDestroy(h2d);
    if (d2h) (void)hipStreamDestroy(d2h);
    for (int r = 0; r < kOutRing; r++)
      for (void *p : {(void *)d_bits[r], (void *)d_iters[r], d_post[r]})
        if (p) (void)hipFree(p);
  }
};

// small-batch path (latency.hip.h): graph tables in the order that path wants, per-XCD codeword state
struct DeviceDecoder::LatencyPath {
  // sliced-ELLPACK tables (latency.hip.h), built in create(), uploaded at first use
  std::vector<uint32_t> h_rslice_ptr, h_rdeg, h_col, h_vslice_ptr, h_vdeg, h_vedge, h_perm, h_inv;
  bool uploaded = false;
  uint32_t *d_rslice_ptr = nullptr, *d_rdeg = nullptr, *d_col = nullptr, *d_vslice_ptr = nullptr, *d_vdeg = nullptr,
           *d_vedge = nullptr, *d_perm = nullptr, *d_inv = nullptr;
  dev::LatencyState slots{};  // 8 slots of {chan, post, msg, rawhard} in one allocation
  dev::LatencySync *d_sync = nullptr;
  uint32_t grid = 0;  // workgroups of the persistent launch (0 = not yet sized from the device's occupancy)
  // pinned host memory the kernel reads and writes itself (sized by the largest call so far): the caller's
  // input; [error word | bits | iterations | posterior]
  char *h_in = nullptr, *h_out = nullptr;
  size_t h_in_bytes = 0, h_out_bytes = 0;

  int pinned(char **p, size_t *have, size_t need) {
    if (*have >= need) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *have = 0;
    const size_t bytes = (need + (size_t(1) << 20) - 1) >> 20 << 20;
    if (hipHostMalloc(reinterpret_cast<void **>(p), bytes, hipHostMallocDefault) != hipSuccess) return -1;
    *have = bytes;
    return 0;
  }
  void release() {
    for (void *p : {(void *)d_rslice_ptr, (void *)d_rdeg, (void *)d_col, (void *)d_vslice_ptr, (void *)d_vdeg, (void *)d_vedge,
                    (void *)d_perm, (void *)d_inv, (void *)slots.base, (void *)d_sync})
      if (p) (void)hipFree(p);
    if (h_in) (void)hipHostFree(h_in);
    if (h_out) (void)hipHostFree(h_out);
  }
};

// small-batch path with the lanes across a codeword's edges (latency_edge.hip.h): the rows packed into wavefront
// chunks, level after level (layered) or all at once (flooding, plus the variables' edge lists)
struct DeviceDecoder::EdgeLatencyPath {
  std::vector<uint32_t> h_level_chunk, h_lane_var, h_lane_info, h_var_ptr, h_var_lane;
  bool uploaded = false, layered = true;
  uint32_t *d_level_chunk = nullptr, *d_lane_var = nullptr, *d_lane_info = nullptr, *d_var_ptr = nullptr, *d_var_lane = nullptr;
  uint32_t n_chunks = 0, grid = 0;
  dev::EdgeLatState slots{};
  dev::LatencySync *d_sync = nullptr;
  char *h_in = nullptr, *h_out = nullptr;  // pinned: the caller's input; [error word | bits | iterations | posterior]
  size_t h_in_bytes = 0, h_out_bytes = 0;

  static int pinned(char **p, size_t *have, size_t need) {
    if (*have >= need) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *have = 0;
    const size_t bytes = (need + (size_t(1) << 16) - 1) >> 16 << 16;
    if (hipHostMalloc(reinterpret_cast<void **>(p), bytes, hipHostMallocDefault) != hipSuccess) return -1;
    *have = bytes;
    return 0;
  }
  void release() {
    for (void *p : {(void *)d_level_chunk, (void *)d_lane_var, (void *)d_lane_info, (void *)d_var_ptr, (void *)d_var_lane,
                    (void *)slots.base, (void *)slots.flags, (void *)d_sync})
      if (p) (void)hipFree(p);
    if (h_in) (void)hipHostFree(h_in);
    if (h_out) (void)hipHostFree(h_out);
  }
};

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) {                            \
      fail(#expr, _e);                                 \
      return -2;                                       \
    }                                                  \
  } while (0)

// the batch entries' straggler pool (device_decoder.h, "pooling"): device buffers of decode_device_pooled
struct DeviceDecoder::StragglerPool {
  uint32_t *d_idx = nullptr, *d_stats = nullptr;  // stats: [converged, failed at the full budget, stragglers] + u64 iterations of the converged
  size_t idx_cap = 0;
  void *d_llrs = nullptr, *d_post = nullptr;
  uint8_t *d_bits = nullptr;
  int32_t *d_its = nullptr, *d_its_all = nullptr;
  size_t llr_bytes = 0, post_bytes = 0, bits_bytes = 0, its_rows = 0, its_all = 0;
  void release() {
    for (void *p : {(void *)d_idx, (void *)d_stats, d_llrs, d_post, (void *)d_bits, (void *)d_its, (void *)d_its_all})
      if (p) (void)hipFree(p);
    *this = StragglerPool();
  }
};

// ---- launch helpers ----------------------------------------------------------------------

struct Tiling {
  uint32_t blocks, threads;
  dev::Sched sched;
};

// Waves are tile-major: wave w works on codeword slice w / wpc and starts at node w % wpc
// (stride wpc).  wpc is rounded so that a slice's waves fill whole workgroups.
inline Tiling make_tiling(uint32_t G, uint32_t tile, uint32_t slice, uint32_t nodes, uint32_t threads,
                   uint32_t target_waves) {
  Tiling t;
  t.threads = threads;
  t.sched.tile = tile;
  t.sched.nchunks = G / slice;
  const uint32_t wpb = threads / 64;
  uint32_t wpc = std::max<uint32_t>(1, target_waves / t.sched.nchunks);
  wpc = std::min<uint32_t>(wpc, std::max<uint32_t>(nodes, 1));
  t.sched.slices_per_tile = std::max<uint32_t>(1, tile / slice);
  // a tile's waves (wpc * slices_per_tile) fill whole workgroups
  while ((uint64_t(wpc) * t.sched.slices_per_tile) % wpb != 0) wpc++;
  t.sched.waves_per_chunk = wpc;
  t.sched.reverse = 0;
  t.sched.per_tile_div = dev::fast_div(wpc * t.sched.slices_per_tile);
  t.sched.spt_div = dev::fast_div(t.sched.slices_per_tile);
  t.sched.tile_div = dev::fast_div(tile);
  t.sched.n_tiles = (t.sched.nchunks + t.sched.slices_per_tile - 1) / t.sched.slices_per_tile;
  t.blocks = static_cast<uint32_t>(uint64_t(wpc) * t.sched.nchunks / wpb);
  return t;
}

// per-call launch tunables (never affect results)
struct Knobs {
  uint32_t rec_dbg = 0;
  bool rec_long = true;  // some row has more than 8 edges
  bool fast = false;  // "@fast" implementation: the approximate Tanh / Phi rule variants
  void *row_scratch = nullptr;  // non-null: the LDS-staged kernels keep their columns there (rows beyond the LDS)
};
inline thread_local Knobs g_knobs;  // set at the top of run_group for the launches of this call
inline thread_local bool t_flood_pace = false;  // set by decode_device for the groups it starts: a one-lane call on the device-resident entry
inline thread_local uint32_t t_pace_lead = 0;  // set by run_any for the group it starts: iterations a paced host runs ahead (0: by schedule)

// LDS-staged kernels: largest block whose [arrays][dmax][threads] columns fit the CU's LDS
inline bool staged_block(uint32_t arrays, uint32_t dmax, size_t elem, uint32_t *threads, size_t *lds) {
  for (uint32_t t : {256u, 128u, 64u}) {
    const size_t bytes = size_t(arrays) * std::max<uint32_t>(dmax, 1) * t * elem;
    if (bytes <= 64 * 1024 || (t == 64 && bytes <= 160 * 1024)) {
      *threads = t;
      *lds = bytes;
      return true;
    }
  }
  return false;
}


// Rows beyond that: the launch keeps its two columns per wavefront in HBM.  A launch of at most kScratchWaves wavefronts
// (make_tiling rounds a slice's waves up to whole workgroups: the allocation follows the tiling actually used).
constexpr uint32_t kScratchWaves = 2048, kScratchThreads = 256;
inline size_t scratch_bytes_for(const Tiling &t, uint32_t dmax, size_t elem) {
  return size_t(t.blocks) * (t.threads / 64) * 2 * dmax * 64 * elem;
}


// Host view of a group's progress word (kernels.hip.h, State::publish).  finished(it) is asked
// before iteration `it` is enqueued: true when every codeword of the group has finished, so that
// all further launches would return at once.  With `throttle` the host also waits until the device
// is within `lead` iterations -- for small groups the launches are so short that an un-throttled
// host would have enqueued most of max_iterations before the first result is known.
struct ProgressPoll {
  const uint64_t *flag;
  uint32_t epoch;
  bool throttle;
  uint32_t lead;
  hipStream_t stream;
  // a paced call that sees no progress at all for this long stops pacing itself (the rest of the group is enqueued at
  // once, as an unpaced call's is): a caller's stream may be gated behind something the calling thread only releases
  // after the call returns (hipStreamWaitValue, a host callback), and then nothing would ever be published
  static constexpr int64_t kStallNs = 200 * 1000 * 1000;
  mutable bool gave_up = false;

  static uint64_t load(const uint64_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
  // codewords of this group still running, as last published (`all` while nothing of this group has been published)
  uint32_t running(uint32_t all) const {
    if (!flag) return all;
    const uint64_t f = load(flag);
    return (f >> 40) == uint64_t(epoch & 0xFFFFFFu) ? static_cast<uint32_t>(f & 0xFFFFFu) : all;
  }
  bool finished(uint32_t it) const {
    if (!flag) return false;
    const uint64_t mine = uint64_t(epoch & 0xFFFFFFu);
    uint64_t f = load(flag);
    if (throttle && !gave_up && it > lead) {
      uint64_t last = f;
      auto since = std::chrono::steady_clock::now();
      for (uint32_t spins = 1;; spins++) {
        if ((f >> 40) == mine && ((f & 0xFFFFFu) == 0 || ((f >> 20) & 0xFFFFFu) + lead >= it)) break;
        if ((spins & 0x3FFu) == 0) {
          if (hipStreamQuery(stream) != hipErrorNotReady) {
            f = load(flag);  // the stream has drained (or failed): nothing more will be published
            break;
          }
          const auto now = std::chrono::steady_clock::now();
          if (f != last) {
            last = f;
            since = now;
          } else if (std::chrono::duration_cast<std::chrono::nanoseconds>(now - since).count() > kStallNs) {
            gave_up = true;
            break;
          }
        }
        f = load(flag);
      }
    }
    return (f >> 40) == mine && (f & 0xFFFFFu) == 0;
  }
};


// Codewords per lane (1, 2 or 4) of the streaming kernels: a wave covers 64 * vec codewords, and
// those slices must tile the layout tile exactly (a 192-codeword tile takes vec = 1: with 128-wide
// slices its last 64 codewords would belong to no wave).
inline uint32_t pick_vec_for(uint32_t tile, uint32_t max_vec, uint32_t wanted) {
  uint32_t vec = std::min<uint32_t>(std::min(max_vec, std::max<uint32_t>(wanted, 1)), 4);
  if (vec == 3) vec = 2;
  while (vec > 1 && tile % (64 * vec) != 0) vec /= 2;
  return vec;
}

// Launchers of the group kernels that are not templates (kernels_group.hip.h): each is defined -- and its kernel compiled --
// once, in device_decoder.hip; the translation units of the schedules call these.
namespace grp {
void init_group(hipStream_t s, uint32_t *done, int32_t *iters, uint32_t *unsat0, uint32_t *unsat1, uint32_t *n_active,
                uint32_t *n_slots, uint32_t *slot_cw, uint32_t nb, uint32_t G);
void latch(hipStream_t s, uint32_t *done, int32_t *iters, uint32_t *unsat, uint32_t *n_active, int32_t iteration, uint32_t G);
void syndrome_bits(hipStream_t s, uint32_t threads, const uint32_t *row_ptr, const uint32_t *edge_col, uint32_t n_rows,
                   const uint64_t *bits, uint32_t *unsat, const uint32_t *n_active, const uint32_t *n_slots, uint32_t W,
                   uint32_t rows_per_thread);
void compact_plan(hipStream_t s, dev::State st, dev::CompactPlan *plan, uint32_t *movers, uint32_t *holes, uint32_t *fill_cw,
                  uint32_t remaining_iterations, dev::CompactRule rule);
void compact_commit(hipStream_t s, dev::State st, const dev::CompactPlan *plan, uint32_t *unsat0, uint32_t *unsat1,
                    uint32_t *n_slots, const uint32_t *fill_cw, uint32_t G);
}  // namespace grp

}  // namespace ldpc
