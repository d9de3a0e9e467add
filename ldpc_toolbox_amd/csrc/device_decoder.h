// Batched belief-propagation decoder on one MI355X (gfx950) -- host-side handle.
//
// This is the device replacement for the reference's per-codeword decoder objects
// (flooding::Decoder<A>, /root/reference/src/decoder/flooding.rs:13-125, and
// horizontal_layered::Decoder<A>, /root/reference/src/decoder/horizontal_layered.rs:17-110)
// generic over the float DecoderArithmetic rules (arithmetic.rs:140-580, 899-1072).
// One handle = one Tanner graph + one rule + one schedule on one device, decoding
// batches of codewords with per-codeword semantics identical to the scalar decode:
// pre-check on the raw input (iterations 0), freeze at the first zero syndrome,
// failure after max_iterations.
//
// HBM layout: every per-codeword array is [row][G] with the
// codeword index fastest, G = codewords per group; a wavefront therefore reads a
// contiguous 256 B..1 KiB row segment for every graph edge it touches.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstddef>
#include <functional>
#include <cstdint>
#include <string>
#include <vector>

#include "implementation.h"
#include "sparse.h"

namespace ldpc {

struct KernelStat {
  uint64_t launches = 0;
  double total_ms = 0.0;
};

// which launches are bracketed with hipEvents when profiling is on
enum KernelKind { kKernelCheck = 0, kKernelVar = 1, kKernelLayer = 2, kKernelOther = 3, kKernelKinds = 4 };

class DeviceDecoder {
 public:
  // puncturing: empty = none, else the reference's block pattern (puncturing.rs:27-40).
  static DeviceDecoder *create(const SparseMatrix &h, const Implementation &impl,
                               const std::vector<uint8_t> &puncturing, int device, std::string *err);
  ~DeviceDecoder();

  size_t n() const { return n_; }
  size_t m() const { return m_; }
  size_t edges() const { return e_; }
  // length of the LLR vector a caller must supply (punctured length if puncturing is set)
  size_t input_len() const { return input_len_; }
  const Implementation &implementation() const { return impl_; }
  int device() const { return device_; }
  uint32_t max_check_degree() const { return max_row_weight_; }
  uint32_t max_variable_degree() const { return max_col_weight_; }
  size_t layers() const { return level_ptr_.empty() ? 0 : level_ptr_.size() - 1; }
  // how the last decode_device / decode_host call was laid out: execution lanes used, codewords per group
  uint32_t last_lanes() const { return last_lanes_; }
  uint64_t last_pooled() const { return last_pooled_; }   // frames of the last call that went through the straggler pool
  size_t last_group() const { return last_group_; }
  // slice width (32 / 64 codewords) of the slice-persistent layered kernel in the last call, 0 = per-level launches
  uint32_t last_persist() const { return last_persist_; }
  // words per check-row record when the flooding min-sum path keeps row records (kernels.hip.h,
  // cn_minsum_rec_kernel), 0 when it keeps per-edge messages
  uint32_t row_records() const { return (rec_ready_ && records_wanted() && opt_lfree_ && !opt_staged_minsum_) ? rec_w_ : 0; }

  // codewords per group (rounded up to the wave tile).  0 = automatic.
  void set_group_size(size_t g) { group_pref_ = g; }
  size_t group_size() const { return group_pref_; }
  // a call of fewer codewords still runs in groups of at least this many (0 = off): a caller that alternates between
  // large and small calls keeps one workspace instead of re-allocating it every time (the simulator's straggler pool)
  void set_min_group(size_t g) { min_group_ = g; }
  // codewords per group a call of `batch` codewords is cut into (the set value, else a default that grows for small graphs)
  size_t preferred_group(size_t batch) const { return pick_group(batch); }
  // The 26 options of set_option (round 6; with the four experiment-only keys 30 where there were 54: the tuning knobs whose alternatives had all been measured within a
  // percent are constants now, see kStreamBlock ... below).  Results never depend on any of them; each selects between
  // forms that tests/ compare bit for bit.  returns false for an unknown key.
  //   which kernels run   "lfree" (0: plain flooding min-sum kernels), "records" (0 / 1 / 2: per-edge messages / row records
  //                       where the graph suits them / wherever possible), "rec_quiet" (0: L-free posteriors stored every
  //                       iteration), "vn_event" (0: the first convergences' rebuild as a launch of its own), "rec_long"
  //                       (1: the record kernel's long-row variant whatever the graph), "staged_minsum" (1: Minsum through
  //                       the generic LDS-staged kernel), "cn_reg" / "hl_reg" / "hl_records" (0: the LDS-staged / two-pass /
  //                       per-edge forms instead of register-resident rows and layered row records), "serial_levels"
  //                       (layered: more dependency levels than this -> row-serial mode), "latency" / "latency_edge"
  //                       (largest batch the single-launch small-batch paths take; 0 = never)
  //   launch geometry     "waves" (target wavefronts per launch), "vec" (codewords per lane: 1, 2, 4), "tile" (codewords per
  //                       layout tile), "rec_run" (consecutive rows per wavefront step of the record kernel)
  //   early termination   "compact" (0: no batch compaction), "compact_first", "compact_every" (checkpoint schedule)
  //   host side           "lanes" (1 or 2 execution lanes; 0 = automatic), "lane_threads", "lane_pace", "lead", "poll" (0: the
  //                       host ignores the progress word), "throttle" (a call on the caller's stream may pace itself),
  //                       "pooling" (straggler pooling inside the batch entries: below)
  // plus "group_size" and "profiling" at the C ABI.  LDPC_TOOLBOX_{GROUP, WAVES, VEC, STAGED_MINSUM} set four of them at
  // construction.  (Builds made with -DLDPC_EXPERIMENTS additionally know "rec_dbg", "lat_debug" -- timing experiments that
  // skip stores or gathers and give WRONG results --, "hl_persist" and "hl_slice"; the product refuses them.)
  bool set_option(const std::string &key, int64_t value);
  void set_profiling(bool on);
  KernelStat kernel_stat(int kind);
  void reset_kernel_stats();

  // Device-resident batch decode.  All pointers are device pointers on device();
  // llrs: [batch][input_len()] f32 (llrs_f64 = false) or f64, one row per codeword;
  // bits: [batch][out_len] u8, first out_len hard decisions of every codeword;
  // iterations: [batch] i32, -1 = failed (may be null);
  // posterior: [batch][n] in the precision of `llrs` (may be null).
  // stream: launch stream (nullptr = the handle's own stream, ordered after everything queued on the
  // legacy default stream at the time of the call, and synchronised on return).
  // returns 0, or a negative error code (message via last_error()).
  int decode_device(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations,
                    uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior,
                    hipStream_t stream);

  // Continuous batching (the reference's workers produce frames until the stop rule fires, ber.rs:297-368): decodes
  // a stream of `total` codewords whose LLR rows are produced on demand, keeping every slot of one group busy -- a
  // slot whose codeword has finished is handed the next codeword of the stream at the next harvest (every few
  // iterations), so the chip stays full until the stream ends.  Per codeword the result is exactly that of the batch
  // entries.  source(first_count, dst, stream) must enqueue, on `stream`, kernels that read the device words
  // first_count[0] (index of the first codeword wanted) and first_count[1] (how many, at most stream_group()) and
  // write their LLR rows [count][input_len()] f32 to dst.  bits [total][out_len], iterations [total]: device memory.
  // Flooding Minsumf32 with row records only (stream_capable()); synchronous.
  bool stream_capable() const;
  size_t stream_group() const { return 4096; }
  uint64_t last_stream_iterations() const { return last_stream_iterations_; }  // group iterations of the last decode_stream
  int decode_stream(const std::function<void(const uint64_t *first_count, float *dst, hipStream_t stream)> &source,
                    float *staging, size_t total, uint32_t max_iterations, uint8_t *bits, size_t out_len,
                    int32_t *iterations);

  // Same contract with host pointers: staged through device buffers group by group.
  int decode_host(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations,
                  uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior);

  // Syndrome of hard decisions (the reference's check_llrs, decoder.rs:157-164, with the parities
  // returned): bits [batch][n] one byte per bit; syndrome [batch][m] (1 = unsatisfied; may be
  // null); weight [batch] (may be null).  Device pointers / host pointers.
  int syndrome_device(const uint8_t *bits, size_t batch, uint8_t *syndrome, uint32_t *weight, hipStream_t stream);
  int syndrome_host(const uint8_t *bits, size_t batch, uint8_t *syndrome, uint32_t *weight);

  const std::string &last_error() const { return error_; }

 private:
  DeviceDecoder() = default;
  struct Workspace;
  int run_group_i8(Workspace &w, const void *llrs, bool llrs_f64, size_t nb, uint32_t max_iterations, uint8_t *bits,
                   size_t out_len, int32_t *iterations, void *posterior, hipStream_t stream, bool may_block);
  template <typename T>
  int run_group(Workspace &w, const void *llrs, bool llrs_f64, size_t nb, uint32_t max_iterations, uint8_t *bits,
                size_t out_len, int32_t *iterations, void *posterior, hipStream_t stream, bool may_block);
  int run_any(Workspace &w, const void *llrs, bool llrs_f64, size_t nb, uint32_t max_iterations, uint8_t *bits,
              size_t out_len, int32_t *iterations, void *posterior, hipStream_t stream, bool may_block,
              bool own_thread = false);
  int ensure_workspace(Workspace &w, size_t group, void *place = nullptr, size_t *need = nullptr);
  int ensure_lanes(uint32_t lanes, size_t group);
  void release_joint();
  int ensure_host_staging(Workspace &w, size_t G, size_t in_elem);
  int ensure_row_scratch(Workspace &w, size_t bytes);
  // host-pointer entry: pinned staging rings, copy streams, batch-wide device output buffers
  struct HostPipe;
  HostPipe *pipe_ = nullptr;
  int ensure_pipe(size_t group, size_t out_len, size_t in_elem, bool posterior);
  int stage_in(const char *src, char *dst, size_t bytes);
  int drain_out(char *dst, const char *src, size_t bytes);
  // recorded by run_group right after the ingest launch (the group's input buffer is free again)
  // small-batch (latency) path: lanes across the rows / variables of one codeword, one persistent launch
  // per call (latency.hip.h).  Flooding Minsumf32 only; the other implementations take the batch kernels.
  struct LatencyPath;
  LatencyPath *lat_ = nullptr;
  // the lane-per-edge small-batch path (latency_edge.hip.h): the layered schedule and, for every rule but Minsumf32,
  // the flooding schedule; f32 and f64 arithmetic
  struct EdgeLatencyPath;
  EdgeLatencyPath *lat_edge_ = nullptr;
  // largest batch that takes it: up to 8 codewords decode one per XCD, larger calls in bundles of up to 8 per XCD that
  // share every phase and barrier; measured against the batched kernels (tools/scalar_probe_layered.py,
  // profiles/r03_latency.txt); the A-Min* rule's serial fold is repeated by every lane of a row: a lower limit
  size_t edge_latency_limit() const;
  uint32_t opt_latency_edge_ = 256;  // "latency_edge": cap on that limit
  size_t edge_lanes_ = 0;           // lane slots of the edge path's message array
  int decode_latency_edge(const void *llrs, bool llrs_f64, bool host_pointers, size_t batch, uint32_t max_iterations,
                             uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior, hipStream_t stream);
  // "latency": largest batch that takes this path (0 = never).  8 codewords decode at once (one per XCD), more
  // take turns; measured against the batched kernels (tools/scalar_probe.py): ahead up to 32 (DVB-S2 1/2 at 2 dB:
  // 16 frames 0.57 vs 2.05 ms, 32 frames 1.11 vs 2.37 ms), level at 64
  uint32_t opt_latency_ = 32;
  static constexpr int kLatencyRetry = -100;  // decode_latency: redo the call with the batched kernels
  static constexpr uint32_t opt_serial_levels_default() { return 512; }
  [[maybe_unused]] uint32_t opt_lat_debug_ = 0;  // "lat_debug" (-DLDPC_EXPERIMENTS builds only): timing probes of the small-batch kernel
  int decode_latency(const void *llrs, bool llrs_f64, bool host_pointers, size_t batch, uint32_t max_iterations,
                     uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior, hipStream_t stream);
  uint32_t lane_count() const;
  bool split_pays(size_t batch) const;
  size_t pick_group(size_t batch) const;
  bool fail(const std::string &msg, hipError_t e = hipSuccess);
  void timed_begin(int kind, hipStream_t s);
  void timed_end(int kind, hipStream_t s);
  void drain_events();

  Implementation impl_;
  int device_ = 0;
  size_t n_ = 0, m_ = 0, e_ = 0, input_len_ = 0;
  uint32_t max_row_weight_ = 0, max_col_weight_ = 0;
  size_t group_pref_ = 0, min_group_ = 0;
  uint32_t opt_tile_ = 0;
  uint32_t opt_waves_ = 0, opt_vec_ = 4;
  bool opt_staged_minsum_ = false;
  // Launch constants that were run-time options up to round 5.  Every one of them had been measured within a percent of its
  // alternatives ("What did not pay"; profiles/r0N_*), each alternative was a set of template variants or
  // a branch nobody took: fixed in round 6 (54 keys in set_option -> 30 with the new "pooling", 641 kernels -> 450 with its three).
  static constexpr uint32_t kStreamBlock = 256;      // threads per workgroup of the streaming kernels (64 / 128: within 1 %)
  static constexpr uint32_t kVnWaves = 128 * 1024;   // wavefronts of a variable-node launch
  static constexpr uint32_t kPackWaves = 16 * 1024;  // ... of the hard-decision packing launch (256 K: -0.7 % on config 3)
  static constexpr uint32_t kSyndThreads = 512 * 1024, kMoveWaves = 64 * 1024, kRetireBlocks = 256;
  // decision rule of the compaction checkpoints (kernels_group.hip.h, CompactRule): waiting until half of the slots are
  // free beats re-packing at a quarter, the cost constants hardly matter (round 2's sweep over Eb/N0)
  static constexpr uint32_t kCompactHorizon = 8, kCompactCostLive = 9, kCompactCostSlots = 0, kCompactMinFreedQ = 2;
  std::string error_;

  // graph tables in HBM
  uint32_t *d_row_ptr_ = nullptr, *d_edge_col_ = nullptr, *d_col_ptr_ = nullptr, *d_col_edge_ = nullptr;
  // layered schedule: rows grouped into dependency levels (hard part 5)
  uint32_t *d_level_rows_ = nullptr;
  uint32_t *d_level_recs_ = nullptr;  // row records of the register-resident level kernels (slice_tasks.h, build_level_recs)
  uint32_t *d_row_recs_ = nullptr;     // flooding: every row's record in row order (cn_reg_kernel)
  uint32_t *d_serial_recs_ = nullptr;  // the same for the row-serial launch: all rows as one level, one record size
  // L-free variables (degree <= 2) of the flooding min-sum path: per-edge aux word, the variables
  // the variable-node kernel still handles ("keep") and the L-free ones ("free"), as compacted CSC
  uint32_t *d_edge_aux_ = nullptr, *d_keep_var_ = nullptr, *d_keep_ptr_ = nullptr, *d_keep_edge_ = nullptr,
           *d_free_var_ = nullptr, *d_free_ptr_ = nullptr, *d_free_edge_ = nullptr;
  uint32_t n_keep_ = 0, n_free_ = 0;
  uint32_t post_rows_keep_ = 0;  // 1 + index of the last variable of degree != 1, 2 (a compaction moves only those posterior rows)
  // row records of the flooding min-sum path (kernels.hip.h, cn_minsum_rec_kernel): per-edge peer word, the
  // (row, slot) pairs of the L-free variables' edges, words per record (3, or 4 for rows too long for the packed form)
  uint32_t *d_edge_peer_ = nullptr, *d_free_rs_ = nullptr, *d_keep_pos_ = nullptr;
  bool opt_rec_long_ = false;  // "rec_long": take the record kernel's long-row variant whatever the graph (A/B)
  bool opt_rec_quiet_ = true;  // "rec_quiet": L-free posteriors are stored only once a slice has a converged codeword
  bool opt_vn_event_ = true;  // "vn_event": the first convergences' L-free posteriors rebuilt inside the variable-node launch (0: a launch of their own)
  uint32_t rec_w_ = 0;
  bool rec_ready_ = false, rec_prefers_ = false;
  // "records": 0 = never, 1 = where the graph suits them (rec_prefers_: the default), 2 = wherever they are possible
  uint32_t opt_records_ = 1;
  bool records_wanted() const { return opt_records_ >= 2 || (opt_records_ == 1 && rec_prefers_); }
  uint32_t opt_rec_run_ = 8;
  static constexpr uint32_t kStreamEvents = 8, kStreamAhead = 4;
  hipEvent_t stream_events_[kStreamEvents] = {};
  static constexpr uint32_t kStreamHarvest = 2;  // iterations between two harvests of decode_stream (experiment builds)
  uint64_t last_stream_iterations_ = 0;
  uint32_t opt_rec_dbg_ = 0;  // "rec_dbg": timing experiments of the record kernel (skips stores / gathers: wrong results)
  bool opt_compact_ = true;
  // schedule of the compaction checkpoints ("compact_first", "compact_every": 0 = 6 and 2 for flooding, 3 and 1 for the
  // layered schedule)
  uint32_t opt_compact_first_ = 0, opt_compact_every_ = 0;
  uint32_t opt_serial_levels_ = 512;  // layered: more dependency levels than this -> row-serial mode
  uint32_t opt_cn_reg_ = 1;  // flooding LDS-staged rules: register-resident rows with row records (0 = cn_staged_kernel)
  uint32_t opt_hl_reg_ = 1;  // layered min-sum: register-resident rows (0 = two-pass form)
  // "lane_threads": the layered schedule's two execution lanes are enqueued by two host threads;
  // "throttle": a call on the CALLER's stream may also wait on the group's progress word between iterations (it then
  // returns when the group is within two iterations of its end instead of at once; the simulator sets it)
  bool opt_lane_threads_ = true, opt_throttle_ = false;
  uint32_t opt_lead_ = 0;  // iterations a paced host may run ahead of its group (0: 1 for a lane's own thread, else 2 layered, 8 flooding)
  bool opt_lane_pace_ = true;  // a lane's own enqueuing thread paces itself on the group's progress word
  bool opt_hl_records_ = true;  // "hl_records": layered min-sum keeps a row's messages as one record (0 = per-edge R)
  std::vector<uint32_t> level_maxdeg_;
  std::vector<uint32_t> level_rec_ptr_;  // [n_levels] first word of a level's records in d_level_recs_
  // slice-persistent layered kernel (kernels.hip.h, hl_slice_kernel; f32 Tanh rule): one launch per iteration; a
  // workgroup owns a slice of 32 or 64 codewords and walks the dependency levels itself.  Task tables for the two slice
  // widths ([0]: 32 codewords, two rows -- or the two halves of a long row -- per wavefront task; [1]: 64, one row).
  // "hl_persist": 0 = one launch per level (default: round 4 measured the persistent form level with it at fixed work and
  // behind it with early termination, profiles/r04_slice_persistent.txt), 1 / 2 = wherever the kernel can run;
  // "hl_slice": 0 = automatic, 32 or 64
  uint32_t *d_slice_tasks_[2] = {nullptr, nullptr}, *d_slice_task_ptr_[2] = {nullptr, nullptr};
  [[maybe_unused]] uint32_t opt_hl_persist_ = 0, opt_hl_slice_ = 0;  // (-DLDPC_EXPERIMENTS builds)
  [[maybe_unused]] bool slice_fits_[2] = {false, false};  // every row of the graph fits a task of that slice width
  uint32_t last_persist_ = 0;  // slice width the last layered group ran with (0: per-level launches)
  bool lfree_ready_ = false, opt_lfree_ = true;
  std::vector<uint32_t> level_ptr_;
  // depuncture map: source block of every pattern block, -1 = punctured
  int32_t *d_src_block_ = nullptr;
  uint32_t pattern_len_ = 0;

  // Two execution lanes (workspace + stream): groups alternate between them, so the idle gaps
  // between one lane's short launches (layered schedule: one per dependency level) are filled by
  // the other's, and the host entry's PCIe copies overlap the other lane's decode.
  Workspace *ws_[2] = {nullptr, nullptr};
  void *joint_slab_ = nullptr;  // both lanes' workspaces (ensure_lanes)
  size_t joint_stride_ = 0, joint_second_ = 0;  // nominal distance of the lanes; the one the placement probe chose
  hipStream_t stream_ = nullptr, stream2_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr, ev_default_ = nullptr;
  int order_after_default_stream(hipStream_t s);
  uint32_t last_lanes_ = 0;
  size_t last_group_ = 0;
  // "pooling" (0 / 1, default 0): STRAGGLER POOLING inside the batch entries (the simulation driver has had it since round 3,
  // csrc/simulator.h).  A call of several chunks learns from its first chunk how many iterations its frames take; later chunks
  // run a reduced budget (2 x average + 8), the frames that have not converged by then are decoded again, together, with the
  // full budget, and their results replace the first pass's.  Per frame the outcome is that of ONE full-budget decode (the
  // decoder starts from the channel LLRs and is deterministic), so outputs do not depend on the option; what it spares is
  // every chunk's nearly empty iterations up to max_iterations behind its few slow or failing frames (the waterfall with
  // the reference's default of 100 iterations, src/cli/ber.rs:64-66).  The host reads each chunk's iteration counts before
  // it starts the next: the call synchronises between chunks (device entry: the library's own stream, or "throttle").
  bool opt_pooling_ = false;
  uint64_t last_pooled_ = 0;
  uint32_t pool_budget_ = 0, pool_budget_max_it_ = 0;  // the budget the last pooled call ended with carries over to the next one at the same limit
  struct StragglerPool;
  StragglerPool *pool_ = nullptr;
  int ensure_pool(size_t batch, size_t rows, size_t out_len, size_t in_elem, bool posterior, bool own_iterations);
  static uint32_t next_pool_budget(double ok, double sum_its_ok, double stragglers, double straggler_its, double failed_full,
                                   uint32_t max_it);
  int decode_device_plain(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits, size_t out_len,
                          int32_t *iterations, void *posterior, hipStream_t stream);
  int decode_device_pooled(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits, size_t out_len,
                           int32_t *iterations, void *posterior, hipStream_t stream);
  int decode_host_plain(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits, size_t out_len,
                        int32_t *iterations, void *posterior);
  int decode_host_pooled(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits, size_t out_len,
                         int32_t *iterations, void *posterior);
  size_t pool_chunk(size_t batch) const;
  uint32_t opt_lanes_ = 0;  // 0 = automatic (2 for the layered schedule, 1 for flooding)
  bool opt_poll_ = true;    // host follows the device's progress word and stops enqueuing a finished group

  bool profiling_ = false;
  struct PendingEvent {
    int kind;
    hipEvent_t a, b;
  };
  std::vector<PendingEvent> pending_;
  std::vector<hipEvent_t> event_pool_;
  KernelStat stats_[kKernelKinds];
};

}  // namespace ldpc
