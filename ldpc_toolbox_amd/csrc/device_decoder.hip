// Host orchestration of the batched HIP decoder: graph upload, workspace in HBM,
// the per-group launch sequence of both schedules.  See device_decoder.h / DESIGN.md.
#include "device_decoder.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <chrono>
#include <thread>

#include "kernels.hip.h"
#include "slice_tasks.h"
#include "kernels_i8.hip.h"
#include "latency.hip.h"
#include "latency_edge.hip.h"

namespace ldpc {

namespace {

uint32_t env_u32(const char *name, uint32_t dflt) {
  const char *s = std::getenv(name);
  if (!s || !*s) return dflt;
  return static_cast<uint32_t>(std::strtoul(s, nullptr, 10));
}

size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

}  // namespace

struct DeviceDecoder::Workspace {
  size_t G = 0;  // codewords per group this workspace is sized for
  size_t elem = 4;
  void *slab = nullptr;  // one allocation; the arrays below are carved from it
  bool borrowed = false;  // the slab is a part of the decoder's joint allocation for both lanes (ensure_lanes)
  size_t slab_bytes = 0;
  size_t pad_kb = 0;
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

bool DeviceDecoder::fail(const std::string &msg, hipError_t e) {
  static std::mutex m;  // (the execution lanes' enqueuing threads may both fail)
  std::lock_guard<std::mutex> lock(m);
  error_ = msg;
  if (e != hipSuccess) error_ += std::string(": ") + hipGetErrorString(e);
  std::fprintf(stderr, "ldpc_toolbox (hip): %s\n", error_.c_str());
  return false;
}

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) {                            \
      fail(#expr, _e);                                 \
      return -2;                                       \
    }                                                  \
  } while (0)

DeviceDecoder *DeviceDecoder::create(const SparseMatrix &h, const Implementation &impl,
                                     const std::vector<uint8_t> &puncturing, int device,
                                     std::string *err) {
  auto bail = [&](const std::string &m) -> DeviceDecoder * {
    if (err) *err = m;
    std::fprintf(stderr, "ldpc_toolbox (hip): %s\n", m.c_str());
    return nullptr;
  };
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return bail("no HIP device available: this library has no CPU decode path");
  if (device < 0 || device >= count) return bail("HIP device index out of range");
  if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice failed");

  SparseMatrix::Csr g = h.csr();
  if (g.n_cols == 0) return bail("parity check matrix has no columns");
  // the row-record kernel fetches a row's first indices as one block: a few entries of slack behind the table
  constexpr uint32_t kTablePad = 16;
  g.edge_col.resize(g.edge_col.size() + kTablePad, 0);
  // degenerate rows the reference panics on at decode time are refused here
  for (uint32_t r = 0; r < g.n_rows; r++) {
    const uint32_t d = g.row_ptr[r + 1] - g.row_ptr[r];
    if (d == 1 && impl.rule != Rule::Phi && impl.rule != Rule::Tanh)
      return bail("check node of degree 1: the " + impl.name + " rule is undefined (arithmetic.rs:513-514)");
    if (d == 0 && impl.rule == Rule::Aminstar)
      return bail("empty check row: the Aminstar rule is undefined (arithmetic.rs:952)");
  }

  DeviceDecoder *d = new DeviceDecoder();
  d->impl_ = impl;
  d->device_ = device;
  d->n_ = g.n_cols;
  d->m_ = g.n_rows;
  d->e_ = g.n_edges;
  d->max_row_weight_ = g.max_row_weight;
  d->max_col_weight_ = g.max_col_weight;
  d->input_len_ = g.n_cols;
  d->group_pref_ = env_u32("LDPC_TOOLBOX_GROUP", 0);
  d->opt_waves_ = env_u32("LDPC_TOOLBOX_WAVES", 0);
  d->opt_unroll_cn_ = env_u32("LDPC_TOOLBOX_UNROLL", 8);
  d->opt_unroll_vn_ = env_u32("LDPC_TOOLBOX_UNROLL_VN", d->opt_unroll_cn_);
  d->opt_vec_ = env_u32("LDPC_TOOLBOX_VEC", 4);
  d->opt_block_ = env_u32("LDPC_TOOLBOX_BLOCK", 256);
  d->opt_staged_minsum_ = env_u32("LDPC_TOOLBOX_STAGED_MINSUM", 0) != 0;

  auto upload = [&](const std::vector<uint32_t> &v, uint32_t **dst) {
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(uint32_t);
    if (hipMalloc(reinterpret_cast<void **>(dst), bytes) != hipSuccess) return false;
    if (!v.empty() && hipMemcpy(*dst, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
      return false;
    return true;
  };
  bool ok = upload(g.row_ptr, &d->d_row_ptr_) && upload(g.edge_col, &d->d_edge_col_) &&
            upload(g.col_ptr, &d->d_col_ptr_) && upload(g.col_edge, &d->d_col_edge_);

  if (ok && impl.schedule == Schedule::Flooding && impl.rule == Rule::Minsum) {
    // L-free variables: degree 1 or 2 (kernels.hip.h, cn_minsum_lfree_kernel)
    std::vector<uint32_t> aux(std::max<uint32_t>(g.n_edges, 1), dev::kAuxNone);
    std::vector<uint32_t> keep_var, keep_ptr{0}, keep_edge, free_var, free_ptr{0}, free_edge;
    for (uint32_t v = 0; v < g.n_cols; v++) {
      const uint32_t s0 = g.col_ptr[v], dv = g.col_ptr[v + 1] - s0;
      const bool is_free = dv == 1 || dv == 2;
      auto &lv = is_free ? free_var : keep_var;
      auto &lp = is_free ? free_ptr : keep_ptr;
      auto &le = is_free ? free_edge : keep_edge;
      lv.push_back(v);
      for (uint32_t j = 0; j < dv; j++) le.push_back(g.col_edge[s0 + j]);
      lp.push_back(static_cast<uint32_t>(le.size()));
      if (dv == 1) aux[g.col_edge[s0]] = dev::kAuxSingle | dev::kAuxWriter;
      if (dv == 2) {
        aux[g.col_edge[s0]] = g.col_edge[s0 + 1] | dev::kAuxWriter;
        aux[g.col_edge[s0 + 1]] = g.col_edge[s0];
      }
    }
#ifdef LDPC_EXPERIMENTS
    if (std::getenv("LDPC_DBG_VNSEQ"))  // timing experiment (wrong results): the variable-node pass reads its messages in order
      for (size_t j = 0; j < keep_edge.size(); j++) keep_edge[j] = static_cast<uint32_t>(j);
#endif
    if (!free_var.empty() && !keep_var.empty() && g.n_edges < dev::kAuxSingle) {
      d->n_keep_ = static_cast<uint32_t>(keep_var.size());
      d->n_free_ = static_cast<uint32_t>(free_var.size());
      d->post_rows_keep_ = keep_var.back() + 1;  // posterior rows up to the last variable the variable-node kernel writes
      ok = upload(aux, &d->d_edge_aux_) && upload(keep_var, &d->d_keep_var_) && upload(keep_ptr, &d->d_keep_ptr_) &&
           upload(keep_edge, &d->d_keep_edge_) && upload(free_var, &d->d_free_var_) &&
           upload(free_ptr, &d->d_free_ptr_) && upload(free_edge, &d->d_free_edge_);
      d->lfree_ready_ = ok;
    }
    // row records (cn_minsum_rec_kernel): where the OTHER message of an L-free variable lives, as (row, slot)
    const uint32_t rec_bits = impl.f64 ? 64u : 32u, rec_packed = impl.f64 ? 58u : 26u;
    if (ok && d->lfree_ready_ && !impl.i8 && g.max_row_weight <= rec_bits && g.n_rows < dev::kPeerSingle) {
      std::vector<uint32_t> rs(std::max<uint32_t>(g.n_edges, 1));  // edge -> row << 6 | slot
      for (uint32_t r = 0; r < g.n_rows; r++)
        for (uint32_t e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++) rs[e] = (r << 6) | (e - g.row_ptr[r]);
      // keep edges: where the variable-node kernel reads the message (its compacted list, variable-major)
      std::vector<uint32_t> peer(std::max<uint32_t>(g.n_edges, 1), dev::kPeerKeep), free_rs(2 * free_var.size(), dev::kAuxNone),
          keep_pos(keep_edge.size());
      for (size_t j = 0; j < keep_edge.size(); j++) {
        peer[keep_edge[j]] = dev::kPeerKeep | static_cast<uint32_t>(j);
        keep_pos[j] = static_cast<uint32_t>(j);
      }
      for (size_t i = 0; i < free_var.size(); i++) {
        const uint32_t v = free_var[i], s0 = g.col_ptr[v], dv = g.col_ptr[v + 1] - s0;
        if (dv == 1) {
          peer[g.col_edge[s0]] = dev::kPeerWriter | (dev::kPeerSingle << 6);
          free_rs[2 * i] = rs[g.col_edge[s0]];
        } else {
          peer[g.col_edge[s0]] = dev::kPeerWriter | rs[g.col_edge[s0 + 1]];
          peer[g.col_edge[s0 + 1]] = rs[g.col_edge[s0]];
          free_rs[2 * i] = rs[g.col_edge[s0]];
          free_rs[2 * i + 1] = rs[g.col_edge[s0 + 1]];
        }
      }
      // the record kernel rebuilds an L-free variable's other message from the peer row's record, which it has at
      // hand only when the peer is the row before or after (staircase codes; a degree-1 variable has no peer).  Codes
      // whose degree-2 variables join distant rows (AR4JA: measured 10 % slower with records) keep per-edge messages.
      size_t far_peers = 0, near_peers = 0;
      for (size_t i = 0; i < free_var.size(); i++) {
        const uint32_t v = free_var[i], s0 = g.col_ptr[v];
        if (g.col_ptr[v + 1] - s0 != 2) continue;
        const uint32_t ra = rs[g.col_edge[s0]] >> 6, rb = rs[g.col_edge[s0 + 1]] >> 6;
        ((ra + 1 == rb || rb + 1 == ra) ? near_peers : far_peers) += 1;
      }
      d->rec_prefers_ = far_peers * 4 <= near_peers + far_peers;  // at most a quarter of the degree-2 variables join distant rows
      d->rec_w_ = g.max_row_weight <= rec_packed ? 3u : 4u;
      peer.resize(peer.size() + kTablePad, dev::kPeerKeep);
      ok = upload(keep_pos, &d->d_keep_pos_);
      ok = ok && upload(peer, &d->d_edge_peer_) && upload(free_rs, &d->d_free_rs_);
      d->rec_ready_ = ok;
    }
  }

  if (ok && impl.schedule == Schedule::Flooding && impl.rule == Rule::Minsum && !impl.f64 && !impl.i8 &&
      g.max_row_weight <= 64 && g.n_rows > 0 && uint64_t(g.max_row_weight) * (g.n_rows + 64) < (1ull << 30) &&
      uint64_t(g.max_col_weight) * (g.n_cols + 64) < (1ull << 30)) {
    // small-batch path: rows in the order of their first variable, 64 to a slice, slot-major inside a slice:
    // edge (position p, slot j) -> id rslice_ptr[p / 64] + j * 64 + p % 64 (messages and `col` share it)
    auto *lp = new LatencyPath();
    std::vector<uint32_t> order(g.n_rows), pos_of_row(g.n_rows), edge_row(std::max<uint32_t>(g.n_edges, 1));
    for (uint32_t r = 0; r < g.n_rows; r++) order[r] = r;
    auto first_var = [&](uint32_t r) { return g.row_ptr[r] < g.row_ptr[r + 1] ? g.edge_col[g.row_ptr[r]] : 0xFFFFFFFFu; };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return first_var(a) < first_var(b); });
    const uint32_t n_rs = (g.n_rows + 63) / 64, n_vs = (g.n_cols + 63) / 64;
    lp->h_rslice_ptr.assign(1, 0);
    lp->h_rdeg.assign(size_t(n_rs) * 64, 0);
    for (uint32_t sl = 0; sl < n_rs; sl++) {
      uint32_t width = 0;
      for (uint32_t p = sl * 64; p < std::min(g.n_rows, sl * 64 + 64); p++) {
        const uint32_t r = order[p], dr = g.row_ptr[r + 1] - g.row_ptr[r];
        pos_of_row[r] = p;
        lp->h_rdeg[p] = dr;
        width = std::max(width, dr);
      }
      lp->h_rslice_ptr.push_back(lp->h_rslice_ptr.back() + width * 64);
    }
    // The variables are renumbered too, in the order of their first appearance when the slots are scanned
    // slot-major over the row positions: neighbouring lanes (rows) then gather neighbouring words of the
    // soft values in EVERY slot where the code has structure -- also in DVB-S2's staircase part, whose
    // natural numbering puts the parity bits of neighbouring positions q words apart (a gather per lane) --
    // and neighbouring variables read neighbouring messages.  The per-codeword arrays (chan, post, rawhard)
    // live in this numbering; only ingest and emit translate (perm / inv).
    lp->h_perm.assign(g.n_cols, 0xFFFFFFFFu);
    {
      uint32_t next = 0;
      for (uint32_t j = 0; j < g.max_row_weight; j++)
        for (uint32_t p = 0; p < g.n_rows; p++) {
          const uint32_t r = order[p];
          if (j < g.row_ptr[r + 1] - g.row_ptr[r]) {
            const uint32_t v = g.edge_col[g.row_ptr[r] + j];
            if (lp->h_perm[v] == 0xFFFFFFFFu) lp->h_perm[v] = next++;
          }
        }
      for (uint32_t v = 0; v < g.n_cols; v++)
        if (lp->h_perm[v] == 0xFFFFFFFFu) lp->h_perm[v] = next++;
    }
    lp->h_inv.assign(g.n_cols, 0);
    for (uint32_t v = 0; v < g.n_cols; v++) lp->h_inv[lp->h_perm[v]] = v;
    lp->h_col.assign(lp->h_rslice_ptr.back() + 8 * 64, 0);  // + padding: a chunk may read past the last slice
    auto edge_id = [&](uint32_t r, uint32_t j) { return lp->h_rslice_ptr[pos_of_row[r] / 64] + j * 64 + pos_of_row[r] % 64; };
    for (uint32_t r = 0; r < g.n_rows; r++)
      for (uint32_t e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++) {
        lp->h_col[edge_id(r, e - g.row_ptr[r])] = lp->h_perm[g.edge_col[e]];
        edge_row[e] = r;
      }
    lp->h_vslice_ptr.assign(1, 0);
    lp->h_vdeg.assign(size_t(n_vs) * 64, 0);
    for (uint32_t sl = 0; sl < n_vs; sl++) {
      uint32_t width = 0;
      for (uint32_t t = sl * 64; t < std::min(g.n_cols, sl * 64 + 64); t++) {
        const uint32_t v = lp->h_inv[t];
        lp->h_vdeg[t] = g.col_ptr[v + 1] - g.col_ptr[v];
        width = std::max(width, lp->h_vdeg[t]);
      }
      lp->h_vslice_ptr.push_back(lp->h_vslice_ptr.back() + width * 64);
    }
    lp->h_vedge.assign(lp->h_vslice_ptr.back() + 8 * 64, 0);
    for (uint32_t t = 0; t < g.n_cols; t++) {
      const uint32_t v = lp->h_inv[t];
      for (uint32_t k = g.col_ptr[v]; k < g.col_ptr[v + 1]; k++) {  // cols[v] order: the reference's sum order
        const uint32_t e = g.col_edge[k], r = edge_row[e];
        lp->h_vedge[lp->h_vslice_ptr[t / 64] + (k - g.col_ptr[v]) * 64 + t % 64] = edge_id(r, e - g.row_ptr[r]);
      }
    }
    d->lat_ = lp;
  }

  if (ok && impl.schedule == Schedule::Flooding && impl.rule == Rule::Tanh && !impl.i8 && d->max_row_weight_ <= kLevelRecShort) {
    // row records of cn_reg_kernel: all rows as one level, in row order
    LevelTables all;
    all.level_ptr = {0u, g.n_rows};
    all.rows.resize(g.n_rows);
    for (uint32_t r = 0; r < g.n_rows; r++) all.rows[r] = r;
    all.maxdeg = {d->max_row_weight_};
    ok = upload(build_level_recs(all, g.row_ptr, g.edge_col).words, &d->d_row_recs_);
  }
  if (ok && impl.schedule == Schedule::Layered) {
    const LevelTables lt = build_levels(g.row_ptr, g.edge_col, g.n_rows, g.n_cols);
    [[maybe_unused]] const uint32_t n_levels = static_cast<uint32_t>(lt.maxdeg.size());
    d->level_ptr_ = lt.level_ptr;
    d->level_maxdeg_ = lt.maxdeg;
    ok = upload(lt.rows, &d->d_level_rows_);
    if (ok) {
      const LevelRecs lr = build_level_recs(lt, g.row_ptr, g.edge_col);
      d->level_rec_ptr_ = lr.rec_ptr;
      ok = upload(lr.words, &d->d_level_recs_);
      if (ok) {
        LevelTables all;
        all.level_ptr = {0u, g.n_rows};
        all.rows = lt.rows;
        all.maxdeg = {lt.maxdeg.empty() ? 0u : *std::max_element(lt.maxdeg.begin(), lt.maxdeg.end())};
        ok = upload(build_level_recs(all, g.row_ptr, g.edge_col).words, &d->d_serial_recs_);
      }
    }
#ifdef LDPC_EXPERIMENTS
    // task tables of the slice-persistent kernel (kernels.hip.h, hl_slice_kernel): the Tanh rule in f32 (a row of its
    // can be shared by two lanes; the other rules keep one launch per level for now)
    if (ok && n_levels <= opt_serial_levels_default() && !impl.i8 && !impl.f64 && impl.rule == Rule::Tanh) {
      for (int k = 0; ok && k < 2; k++) {
        const SliceTasks st = build_slice_tasks(lt, g.row_ptr, g.edge_col, k == 0 ? 2u : 1u, true);
        d->slice_fits_[k] = st.fits;
        ok = upload(st.tasks, &d->d_slice_tasks_[k]) && upload(st.task_ptr, &d->d_slice_task_ptr_[k]);
      }
    }
#endif
  }

  // small-batch path with a lane per edge (latency_edge.hip.h): the rows are packed, whole, into chunks of at most 64
  // lanes (one wavefront) -- level after level for the layered schedule, all rows in order for flooding, which also
  // gets the variables' edge lists (cols[v] order) as lane indices.  Flooding Minsumf32 keeps latency.hip.h's kernel.
  if (ok && !impl.fast && g.max_row_weight <= 64 && g.n_rows > 0 && d->lat_ == nullptr &&
      (impl.schedule == Schedule::Flooding || d->level_ptr_.size() <= size_t(opt_serial_levels_default()) + 1)) {
    auto *lp = new EdgeLatencyPath();
    lp->layered = impl.schedule == Schedule::Layered;
    lp->h_level_chunk.assign(1, 0);
    std::vector<uint32_t> edge_lane(std::max<uint32_t>(g.n_edges, 1), 0);
    uint32_t fill = 0;  // lanes used in the open chunk
    auto close = [&]() {
      if (fill == 0) return;
      const size_t c0 = lp->h_lane_var.size() - fill;
      uint32_t dmax = 0;
      for (size_t k = c0; k < c0 + fill; k++) dmax = std::max(dmax, (lp->h_lane_info[k] >> 8) & 0xFFu);
      lp->h_lane_var.resize(c0 + 64, dev::kNoLane);
      lp->h_lane_info.resize(c0 + 64, 0);
      for (size_t k = c0; k < c0 + 64; k++) lp->h_lane_info[k] |= dmax << 16;
      fill = 0;
    };
    auto add_row = [&](uint32_t r) {
      const uint32_t e0 = g.row_ptr[r], dr = g.row_ptr[r + 1] - e0;
      if (dr == 0) return;  // an empty row has no message and an even parity
      if (fill + dr > 64) close();
      for (uint32_t i = 0; i < dr; i++) {
        edge_lane[e0 + i] = static_cast<uint32_t>(lp->h_lane_var.size());
        lp->h_lane_var.push_back(g.edge_col[e0 + i]);
        lp->h_lane_info.push_back(i | (dr << 8));
      }
      fill += dr;
    };
    if (lp->layered) {
      std::vector<uint32_t> level_rows(g.n_rows);
      if (hipMemcpy(level_rows.data(), d->d_level_rows_, size_t(g.n_rows) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        ok = false;
      for (size_t l = 0; ok && l + 1 < d->level_ptr_.size(); l++) {
        for (uint32_t idx = d->level_ptr_[l]; idx < d->level_ptr_[l + 1]; idx++) add_row(level_rows[idx]);
        close();
        lp->h_level_chunk.push_back(static_cast<uint32_t>(lp->h_lane_var.size() / 64));
      }
    } else {
      for (uint32_t r = 0; r < g.n_rows; r++) add_row(r);
      close();
      lp->h_level_chunk.push_back(static_cast<uint32_t>(lp->h_lane_var.size() / 64));
      lp->h_var_ptr.assign(g.col_ptr.begin(), g.col_ptr.end());
      lp->h_var_lane.resize(std::max<uint32_t>(g.n_edges, 1), 0);
      for (uint32_t j = 0; j < g.n_edges; j++) lp->h_var_lane[j] = edge_lane[g.col_edge[j]];
    }
    lp->n_chunks = static_cast<uint32_t>(lp->h_lane_var.size() / 64);
    d->edge_lanes_ = lp->h_lane_var.size();
    d->lat_edge_ = lp;
  }

  if (ok && !puncturing.empty()) {
    size_t trues = 0;
    for (uint8_t p : puncturing) trues += p ? 1 : 0;
    if (trues == 0 || g.n_cols % puncturing.size() != 0) {
      delete d;
      return bail("codeword size not divisible by puncturing pattern length");
    }
    std::vector<int32_t> src(puncturing.size());
    int32_t j = 0;
    for (size_t k = 0; k < puncturing.size(); k++) src[k] = puncturing[k] ? j++ : -1;
    d->pattern_len_ = static_cast<uint32_t>(puncturing.size());
    d->input_len_ = g.n_cols / puncturing.size() * trues;
    ok = hipMalloc(reinterpret_cast<void **>(&d->d_src_block_), src.size() * sizeof(int32_t)) == hipSuccess &&
         hipMemcpy(d->d_src_block_, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok) ok = hipStreamCreateWithFlags(&d->stream_, hipStreamNonBlocking) == hipSuccess;
  if (ok) ok = hipStreamCreateWithFlags(&d->stream2_, hipStreamNonBlocking) == hipSuccess;
  if (ok) ok = hipEventCreateWithFlags(&d->ev_fork_, hipEventDisableTiming) == hipSuccess;
  if (ok) ok = hipEventCreateWithFlags(&d->ev_join_, hipEventDisableTiming) == hipSuccess;
  if (ok) ok = hipEventCreateWithFlags(&d->ev_skew_, hipEventDisableTiming) == hipSuccess;
  if (ok) ok = hipEventCreateWithFlags(&d->ev_default_, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    delete d;
    return bail("device allocation / upload of the graph tables failed");
  }
  d->ws_[0] = new Workspace();
  d->ws_[1] = new Workspace();
  return d;
}

DeviceDecoder::~DeviceDecoder() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (stream2_) (void)hipStreamSynchronize(stream2_);
  for (auto &p : pending_) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : event_pool_) (void)hipEventDestroy(e);
  if (pipe_) {
    pipe_->release();
    delete pipe_;
  }
  if (lat_) {
    lat_->release();
    delete lat_;
  }
  if (lat_edge_) {
    lat_edge_->release();
    delete lat_edge_;
  }
  for (Workspace *w : ws_)
    if (w) {
      w->release();
      delete w;
    }
  if (joint_slab_) (void)hipFree(joint_slab_);
  for (void *p : {(void *)d_row_ptr_, (void *)d_edge_col_, (void *)d_col_ptr_, (void *)d_col_edge_,
                  (void *)d_level_rows_, (void *)d_level_recs_, (void *)d_serial_recs_, (void *)d_row_recs_, (void *)d_src_block_, (void *)d_edge_aux_, (void *)d_keep_var_,
                  (void *)d_keep_ptr_, (void *)d_keep_edge_, (void *)d_free_var_, (void *)d_free_ptr_,
                  (void *)d_free_edge_, (void *)d_edge_peer_, (void *)d_free_rs_, (void *)d_keep_pos_,
                  (void *)d_slice_tasks_[0], (void *)d_slice_tasks_[1], (void *)d_slice_task_ptr_[0],
                  (void *)d_slice_task_ptr_[1]})
    if (p) (void)hipFree(p);
  for (auto e : stream_events_)
    if (e) (void)hipEventDestroy(e);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  if (ev_skew_) (void)hipEventDestroy(ev_skew_);
  if (ev_default_) (void)hipEventDestroy(ev_default_);
  if (stream_) (void)hipStreamDestroy(stream_);
  if (stream2_) (void)hipStreamDestroy(stream2_);
}

// ---- profiling: hipEvents around the bracketed launches, on the launch stream -----------

void DeviceDecoder::set_profiling(bool on) { profiling_ = on; }

bool DeviceDecoder::set_option(const std::string &key, int64_t value) {
  if (value < 0) return false;
  const uint32_t v = static_cast<uint32_t>(value);
  if (key == "waves")
    opt_waves_ = v;
  else if (key == "unroll_cn")
    opt_unroll_cn_ = v;
  else if (key == "unroll_vn")
    opt_unroll_vn_ = v;
  else if (key == "vec")
    opt_vec_ = v;
  else if (key == "block")
    opt_block_ = v;
  else if (key == "tile")
    opt_tile_ = v;
  else if (key == "lfree")
    opt_lfree_ = v != 0;
  else if (key == "records")
    opt_records_ = v != 0 ? (v >= 2 ? 2 : 1) : 0;  // 2: also where the graph's peers are distant rows
  else if (key == "rec_run")
    opt_rec_run_ = std::max<uint32_t>(v, 1);
  else if (key == "rec_unroll")
    opt_rec_unroll_ = v;
#ifdef LDPC_EXPERIMENTS
  else if (key == "rec_dbg")
    opt_rec_dbg_ = v;
  else if (key == "lat_debug")
    opt_lat_debug_ = v;
#endif
  else if (key == "rec_quiet")
    opt_rec_quiet_ = v != 0;
  else if (key == "vn_event")
    opt_vn_event_ = v != 0;
  else if (key == "waves_pack")
    opt_waves_pack_ = v;
  else if (key == "rec_long")
    opt_rec_long_ = v != 0;
  else if (key == "vn_reverse")
    opt_vn_reverse_ = v != 0;
  else if (key == "stream_harvest")
    opt_stream_harvest_ = std::max<uint32_t>(v, 1);
  else if (key == "compact")
    opt_compact_ = v != 0;
  else if (key == "lfree_unroll")
    opt_lfree_unroll_ = v;
  else if (key == "lfree_nt_in")
    opt_lfree_nt_in_ = v != 0;
  else if (key == "waves_vn")
    opt_waves_vn_ = v;
  else if (key == "nt")
    opt_nt_ = v != 0;
  else if (key == "nt_vn")
    opt_nt_vn_ = v != 0;
  else if (key == "pad_kb")
    opt_pad_kb_ = v;
  else if (key == "staged_minsum")
    opt_staged_minsum_ = v != 0;
  else if (key == "hl_reg")
    opt_hl_reg_ = v;
  else if (key == "cn_reg")
    opt_cn_reg_ = v;
  else if (key == "hl_records")
    opt_hl_records_ = v != 0;
#ifdef LDPC_EXPERIMENTS  // the slice-persistent layered kernel exists in experiment builds only (round 5)
  else if (key == "hl_persist")
    opt_hl_persist_ = std::min<uint32_t>(v, 2);
  else if (key == "hl_slice")
    opt_hl_slice_ = (v == 32 || v == 64) ? v : 0;
#endif
  else if (key == "lane_pad_kb")
    opt_lane_pad_kb_ = v;
  else if (key == "lane_align_mb")
    opt_lane_align_mb_ = v;
  else if (key == "lane_threads")
    opt_lane_threads_ = v != 0;
  else if (key == "host_split")
    opt_host_split_ = v != 0;
  else if (key == "throttle")
    opt_throttle_ = v != 0;
  else if (key == "lead")
    opt_lead_ = v;
  else if (key == "lane_pace")
    opt_lane_pace_ = v != 0;
  else if (key == "lanes")
    opt_lanes_ = std::min<uint32_t>(v, 2);
  else if (key == "poll")
    opt_poll_ = v != 0;
  else if (key == "lane_skew")
    opt_lane_skew_ = v;
  else if (key == "latency") {
    opt_latency_ = v;
    opt_latency_edge_ = v == 0 ? 0 : std::max<uint32_t>(v, 64);  // 0 switches both small-batch paths off
  } else if (key == "latency_edge")
    opt_latency_edge_ = v;
  else if (key == "lat_grid") {
    opt_lat_grid_ = v;
    if (lat_edge_) lat_edge_->grid = 0;  // re-sized at the next call
  }
  else if (key == "compact_horizon")
    opt_compact_horizon_ = v;
  else if (key == "compact_cost_live")
    opt_compact_cost_live_ = v;
  else if (key == "compact_cost_slots")
    opt_compact_cost_slots_ = v;
  else if (key == "compact_min_freed_q")
    opt_compact_min_freed_q_ = v;
  else if (key == "compact_first")
    opt_compact_first_ = v;
  else if (key == "serial_levels")
    opt_serial_levels_ = v;
  else if (key == "synd_threads")
    opt_synd_threads_ = std::max<uint32_t>(v, 1024);
  else if (key == "move_waves")
    opt_move_waves_ = std::max<uint32_t>(v, 64);
  else if (key == "retire_blocks")
    opt_retire_blocks_ = std::max<uint32_t>(v, 1);
  else if (key == "compact_every")
    opt_compact_every_ = v;
  else
    return false;
  return true;
}

void DeviceDecoder::timed_begin(int kind, hipStream_t s) {
  if (!profiling_) return;
  PendingEvent p;
  p.kind = kind;
  auto get = [&]() {
    hipEvent_t e;
    if (!event_pool_.empty()) {
      e = event_pool_.back();
      event_pool_.pop_back();
    } else {
      (void)hipEventCreate(&e);
    }
    return e;
  };
  p.a = get();
  p.b = get();
  (void)hipEventRecord(p.a, s);
  pending_.push_back(p);
}

void DeviceDecoder::timed_end(int, hipStream_t s) {
  if (!profiling_) return;
  (void)hipEventRecord(pending_.back().b, s);
}

void DeviceDecoder::drain_events() {
  for (auto &p : pending_) {
    (void)hipEventSynchronize(p.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      stats_[p.kind].launches++;
      stats_[p.kind].total_ms += ms;
    }
    event_pool_.push_back(p.a);
    event_pool_.push_back(p.b);
  }
  pending_.clear();
}

KernelStat DeviceDecoder::kernel_stat(int kind) {
  drain_events();
  return (kind >= 0 && kind < kKernelKinds) ? stats_[kind] : KernelStat();
}

void DeviceDecoder::reset_kernel_stats() {
  drain_events();
  for (auto &s : stats_) s = KernelStat();
}

// ---- workspace ---------------------------------------------------------------------------

size_t DeviceDecoder::pick_group(size_t batch) const {
  // wave tile: 64 codewords x VEC; the staged kernels use VEC = 1
  // row-serial layered mode (run_group): a group takes the same time whatever its size until the
  // waves fill the chip, so the default group is large there
  const bool serial = impl_.schedule == Schedule::Layered && level_ptr_.size() > size_t(opt_serial_levels_) + 1;
  // Small graphs: a 4096-codeword launch is over in tens of microseconds and the launch sequence shows through (AR4JA
  // r=1/2 k=1024, 10 iterations of Minsumf32: 0.61 of the roofline in groups of 4096, 0.86 in groups of 65536;
  // HLMinsumf32 0.39 -> 0.71): the default group doubles while its messages stay within those of a DVB-S2 group.
  size_t by_size = 4096;
  while (by_size < 65536 && e_ * by_size * 2 <= size_t(226799) * 4096) by_size *= 2;
  size_t g = group_pref_ ? group_pref_ : std::max<size_t>(serial ? 16384 : 4096, by_size);
  g = std::min(g, std::max(round_up(batch, 64), std::min(min_group_, g)));
  g = round_up(g, 64);
  if (impl_.i8) return round_up(g, 256);  // a lane packs four codewords: 256-codeword slices only
  if (g >= 256) g = g / 256 * 256;  // whole float4 tiles for the streaming kernels
  return g;
}

uint32_t DeviceDecoder::lane_count() const {
  if (opt_lanes_) return opt_lanes_;
  if (impl_.schedule == Schedule::Layered) return 2u;
  // flooding: two launches per iteration fill the chip by themselves, but with two half-batches in flight one lane's
  // variable-node pass runs beside the other's check-node pass and each lane's launch tails and dispatch gaps are
  // filled by the other.  Measured at fixed work, 50 iterations (tools/p2_probe.py, tools/lanes_placement.py,
  // profiles/r03_lanes.txt): Tanhf32 +4..+12 % on every code and in every placement; Minsumf32 on 5G NR BG1 Zc=384
  // +4..+10 % in every placement tried, DVB-S2 3/5 +7 %, 9/10 +2.4 % -- but on DVB-S2 1/2, whose check-node launch
  // alone already runs at the fabric's rate, +4.5 % or -2.5 % depending on where the allocator put the two lanes'
  // workspaces (physical placement: no alignment or distance was good in every process), and AR4JA -1 %.  So: Tanh
  // always; min-sum where the rows are long (more than 8 edges: the launches that leave the memory system room);
  // one lane otherwise (Aminstarf32 +2.9 %, Minstarapproxf32 +2.6 %, Phif32 -0.7 %, f64 and 8-bit -2..-4 %: within
  // what placement alone moves).
  if (impl_.f64 || impl_.i8) return 1u;
  if (impl_.rule == Rule::Tanh) return 2u;
  // (with the event brackets of "profiling" around every launch the lanes no longer overlap usefully: one lane)
  if (impl_.rule == Rule::Minsum && max_row_weight_ > 8 && !profiling_) return 2u;
  return 1u;
}

bool DeviceDecoder::split_pays(size_t batch) const {
  return opt_lanes_ == 2 || (batch >= 2048 && n_ * (batch / 2) >= size_t(50) * 1000 * 1000);
}

// place: carve the arrays from this memory (a part of the joint allocation of ensure_lanes) instead of an allocation of
// the workspace's own; need: only report the bytes such a workspace takes
int DeviceDecoder::ensure_workspace(Workspace &w, size_t G, void *place, size_t *need) {
  const size_t elem = impl_.i8 ? 2 : (impl_.f64 ? 8 : 4);
  const bool records = lfree_ready_ && rec_ready_ && records_wanted() && opt_lfree_;
  if (!need) {
    if (w.G == G && w.elem == elem && w.chan && w.pad_kb == opt_pad_kb_ && w.records == records && (!place || w.slab == place))
      return 0;
    w.release();
  }
  const size_t W = G / 64;
  // One slab, carved: the big arrays first, each start 2 MiB-aligned plus a configurable skew.
  // (Separate hipMalloc calls made the check-node kernel's time vary by ~12 % from one
  // allocation to the next; a single slab keeps the relative placement fixed.)
  const size_t align = size_t(2) << 20, skew = size_t(opt_pad_kb_) << 10;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = off;
    off = round_up(off + std::max<size_t>(bytes, 256), align) + skew;
    return at;
  };
  const size_t o_msg = carve(std::max<size_t>(e_, 1) * G * elem);
  const size_t o_msg2 = (lfree_ready_ && !records) ? carve(std::max<size_t>(e_, 1) * G * elem) : 0;
  const size_t rec_bytes = records ? std::max<size_t>(m_, 1) * rec_w_ * G * elem : 0;
  const size_t o_rec0 = records ? carve(rec_bytes) : 0, o_rec1 = records ? carve(rec_bytes) : 0;
  const size_t o_post = carve(n_ * G * elem);
  const size_t o_chan = carve(n_ * G * elem);
  const size_t o_perm = carve(4 * G * sizeof(uint32_t) + 1024);
  const size_t o_raw = carve(n_ * W * sizeof(uint64_t));
  const size_t o_hard = carve(n_ * W * sizeof(uint64_t));
  const size_t o_flags = carve(9 * G * sizeof(uint32_t) + 1024);
  if (need) {
    *need = off;
    return 0;
  }
  w.records = records;
  if (hipHostMalloc(reinterpret_cast<void **>(&w.h_flag), 64, hipHostMallocMapped) == hipSuccess) {
    *w.h_flag = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&w.d_flag), w.h_flag, 0) != hipSuccess) w.d_flag = nullptr;
  } else {
    w.h_flag = nullptr;  // no progress word: the host simply enqueues every iteration
  }
  w.G = G;
  w.elem = elem;
  w.pad_kb = opt_pad_kb_;
  if (place) {
    w.slab = place;
    w.borrowed = true;
  } else {
    HIP_TRY(hipMalloc(&w.slab, off));
  }
  w.slab_bytes = off;
  char *base = static_cast<char *>(w.slab);
  w.msg = base + o_msg;
  w.msg2 = (lfree_ready_ && !records) ? base + o_msg2 : nullptr;
  w.rec[0] = records ? base + o_rec0 : nullptr;
  w.rec[1] = records ? base + o_rec1 : nullptr;
  w.post = base + o_post;
  w.chan = base + o_chan;
  w.perm = reinterpret_cast<uint32_t *>(base + o_perm);
  w.slot_cw = w.perm + G;
  w.slot_tmp = w.perm + 2 * G;
  w.fill_cw = w.perm + 3 * G;
  w.n_slots = w.perm + 4 * G;
  w.plan = reinterpret_cast<dev::CompactPlan *>(w.perm + 4 * G + 16);
  w.rawbits = reinterpret_cast<uint64_t *>(base + o_raw);
  w.hardbits = reinterpret_cast<uint64_t *>(base + o_hard);
  uint32_t *flags = reinterpret_cast<uint32_t *>(base + o_flags);
  w.done = flags;
  w.unsat0 = flags + G;
  w.unsat1 = flags + 2 * G;
  w.iters = reinterpret_cast<int32_t *>(flags + 3 * G);
  w.n_active = flags + 4 * G;
  w.scratch_flags = flags + 4 * G + 64;
  w.slice_state = flags + 6 * G;
  w.it0 = flags + 7 * G;
  w.holes = flags + 8 * G;
  w.stream_plan = reinterpret_cast<dev::StreamPlan *>(flags + 9 * G + 64);
  if (std::getenv("LDPC_TOOLBOX_DEBUG"))
    std::fprintf(stderr, "ldpc_toolbox (hip): workspace G=%zu slab=%p bytes=%zu msg=+%zx post=+%zx chan=+%zx\n", G,
                 w.slab, off, o_msg, o_post, o_chan);
  return 0;
}

// Both lanes' workspaces in ONE allocation, a fixed distance apart.  With two allocations the lanes' relative placement
// changed from one process (or one reallocation) to the next, and with it whether the two lanes' concurrent streams
// collide in the memory system: the same two-lane call took 100.2 ms or 107.0 ms (one lane: 104.1 ms, always) --
// tools/lanes_placement.py, profiles/r03_lanes.txt.
int DeviceDecoder::ensure_lanes(uint32_t lanes, size_t G) {
  if (lanes < 2) {
    Workspace &w = *ws_[0];
    if (w.borrowed && w.G != G) release_joint();
    return ensure_workspace(w, G);
  }
  size_t bytes = 0;
  if (int rc = ensure_workspace(*ws_[0], G, nullptr, &bytes)) return rc;
  const size_t lane_align = size_t(std::max<uint32_t>(opt_lane_align_mb_, 2)) << 20;
  const size_t stride = round_up(bytes, lane_align) + (size_t(opt_lane_pad_kb_) << 10);
  const size_t elem = impl_.i8 ? 2 : (impl_.f64 ? 8 : 4);
  const bool records = lfree_ready_ && rec_ready_ && records_wanted() && opt_lfree_;
  auto current = [&](const Workspace &w, const char *at) {
    return w.borrowed && w.slab == at && w.G == G && w.elem == elem && w.pad_kb == opt_pad_kb_ && w.records == records;
  };
  char *base = joint_slab_ ? reinterpret_cast<char *>(round_up(reinterpret_cast<size_t>(joint_slab_), lane_align)) : nullptr;
  if (base && joint_stride_ == stride && current(*ws_[0], base) && current(*ws_[1], base + joint_second_)) return 0;
  release_joint();
  HIP_TRY(hipMalloc(&joint_slab_, stride + bytes + lane_align));
  base = reinterpret_cast<char *>(round_up(reinterpret_cast<size_t>(joint_slab_), lane_align));
  joint_stride_ = stride;
  joint_second_ = stride;
  if (int rc = ensure_workspace(*ws_[0], G, base)) return rc;
  return ensure_workspace(*ws_[1], G, base + joint_second_);
}

void DeviceDecoder::release_joint() {
  for (Workspace *w : ws_)
    if (w && w->borrowed) w->release();
  if (joint_slab_) (void)hipFree(joint_slab_);
  joint_slab_ = nullptr;
  joint_stride_ = 0;
  joint_second_ = 0;
}

// ---- launch helpers ----------------------------------------------------------------------

namespace {

struct Tiling {
  uint32_t blocks, threads;
  dev::Sched sched;
};

// Waves are tile-major: wave w works on codeword slice w / wpc and starts at node w % wpc
// (stride wpc).  wpc is rounded so that a slice's waves fill whole workgroups.
Tiling make_tiling(uint32_t G, uint32_t tile, uint32_t slice, uint32_t nodes, uint32_t threads,
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
  bool nt = true, nt_vn = true;  // nontemporal message accesses in the check / variable kernels
  bool lfree_nt_in = false;
  uint32_t lfree_unroll = 4, rec_unroll = 4, rec_dbg = 0;
  bool rec_long = true;  // some row has more than 8 edges
  bool fast = false;  // "@fast" implementation: the approximate Tanh / Phi rule variants
  void *row_scratch = nullptr;  // non-null: the LDS-staged kernels keep their columns there (rows beyond the LDS)
};
thread_local Knobs g_knobs;  // set at the top of run_group for the launches of this call
thread_local bool t_flood_pace = false;  // set by decode_device for the groups it starts: a one-lane call on the device-resident entry
thread_local uint32_t t_pace_lead = 0;  // set by run_any for the group it starts: iterations a paced host runs ahead (0: by schedule)

template <typename T>
struct Launch {
  // flooding min-sum check nodes: VEC x mask width x unroll x FIRST
  template <int VEC, typename MASK, bool FIRST>
  static void cn_minsum_u(uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const T *L, T *msg, uint32_t *unsat) {
    if (g_knobs.nt) {
      if (unroll >= 8)
        dev::cn_minsum_kernel<T, VEC, MASK, 8, FIRST, true><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, L, msg, unsat);
      else
        dev::cn_minsum_kernel<T, VEC, MASK, 4, FIRST, true><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, L, msg, unsat);
    } else {
      if (unroll >= 8)
        dev::cn_minsum_kernel<T, VEC, MASK, 8, FIRST, false><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, L, msg, unsat);
      else
        dev::cn_minsum_kernel<T, VEC, MASK, 4, FIRST, false><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, L, msg, unsat);
    }
  }
  template <int VEC, bool FIRST>
  static void cn_minsum_m(bool wide_mask, uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const T *L, T *msg, uint32_t *unsat) {
    if (wide_mask)
      cn_minsum_u<VEC, uint64_t, FIRST>(unroll, t, s, g, st, L, msg, unsat);
    else
      cn_minsum_u<VEC, uint32_t, FIRST>(unroll, t, s, g, st, L, msg, unsat);
  }
  // L-free variant (double-buffered messages)
  template <int VEC, typename MASK, bool FIRST>
  static void cn_lfree_u(const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st, const T *chan,
                         T *post, const T *msg_in, T *msg_out, uint32_t *unsat) {
    if (g_knobs.lfree_unroll >= 8) {
      if (g_knobs.lfree_nt_in)
        dev::cn_minsum_lfree_kernel<T, VEC, MASK, 8, FIRST, true, true><<<t.blocks, t.threads, 0, s>>>(
            g, t.sched, st, chan, post, msg_in, msg_out, unsat);
      else
        dev::cn_minsum_lfree_kernel<T, VEC, MASK, 8, FIRST, true, false><<<t.blocks, t.threads, 0, s>>>(
            g, t.sched, st, chan, post, msg_in, msg_out, unsat);
    } else {
      if (g_knobs.lfree_nt_in)
        dev::cn_minsum_lfree_kernel<T, VEC, MASK, 4, FIRST, true, true><<<t.blocks, t.threads, 0, s>>>(
            g, t.sched, st, chan, post, msg_in, msg_out, unsat);
      else
        dev::cn_minsum_lfree_kernel<T, VEC, MASK, 4, FIRST, true, false><<<t.blocks, t.threads, 0, s>>>(
            g, t.sched, st, chan, post, msg_in, msg_out, unsat);
    }
  }
  template <int VEC, bool FIRST>
  static void cn_lfree_m(bool wide_mask, const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                         const T *chan, T *post, const T *msg_in, T *msg_out, uint32_t *unsat) {
    if (wide_mask)
      cn_lfree_u<VEC, uint64_t, FIRST>(t, s, g, st, chan, post, msg_in, msg_out, unsat);
    else
      cn_lfree_u<VEC, uint32_t, FIRST>(t, s, g, st, chan, post, msg_in, msg_out, unsat);
  }
  template <bool FIRST>
  static void cn_lfree(uint32_t vec, bool wide_mask, const Tiling &t, hipStream_t s, const dev::Graph &g,
                       const dev::State &st, const T *chan, T *post, const T *msg_in, T *msg_out,
                       uint32_t *unsat) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      cn_lfree_m<kMaxVec, FIRST>(wide_mask, t, s, g, st, chan, post, msg_in, msg_out, unsat);
    else if (vec >= 2)
      cn_lfree_m<2, FIRST>(wide_mask, t, s, g, st, chan, post, msg_in, msg_out, unsat);
    else
      cn_lfree_m<1, FIRST>(wide_mask, t, s, g, st, chan, post, msg_in, msg_out, unsat);
  }

  // row records (cn_minsum_rec_kernel): VEC x words per record x loads in flight x FIRST
  template <int VEC, int RECW, bool FIRST>
  static void cn_rec_u(const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st, const T *chan, T *post,
                       const T *rec_in, T *rec_out, T *msg, uint32_t *unsat, uint32_t run) {
    // (rows of at most 8 edges -- DVB-S2 up to rate 1/2, most 5G NR rows are longer -- take the variant without the
    // further-rounds code)
    // (eight loads in flight per lane; the four-load variant of earlier rounds, a tuning knob nothing selected, is gone)
    if (g_knobs.rec_long)
      dev::cn_minsum_rec_kernel<T, VEC, RECW, 8, FIRST, true, false, true><<<t.blocks, t.threads, 0, s>>>(
          g, t.sched, st, chan, post, rec_in, rec_out, msg, unsat, run LDPC_DBG_ARG(g_knobs.rec_dbg));
    else
      dev::cn_minsum_rec_kernel<T, VEC, RECW, 8, FIRST, true, false, false><<<t.blocks, t.threads, 0, s>>>(
          g, t.sched, st, chan, post, rec_in, rec_out, msg, unsat, run LDPC_DBG_ARG(g_knobs.rec_dbg));
  }
  template <int VEC, bool FIRST>
  static void cn_rec_w(uint32_t recw, const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st, const T *chan,
                       T *post, const T *rec_in, T *rec_out, T *msg, uint32_t *unsat, uint32_t run) {
    if (recw == 3)
      cn_rec_u<VEC, 3, FIRST>(t, s, g, st, chan, post, rec_in, rec_out, msg, unsat, run);
    else
      cn_rec_u<VEC, 4, FIRST>(t, s, g, st, chan, post, rec_in, rec_out, msg, unsat, run);
  }
  template <bool FIRST>
  static void cn_rec(uint32_t vec, uint32_t recw, const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                     const T *chan, T *post, const T *rec_in, T *rec_out, T *msg, uint32_t *unsat, uint32_t run) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      cn_rec_w<kMaxVec, FIRST>(recw, t, s, g, st, chan, post, rec_in, rec_out, msg, unsat, run);
    else if (vec >= 2)
      cn_rec_w<2, FIRST>(recw, t, s, g, st, chan, post, rec_in, rec_out, msg, unsat, run);
    else
      cn_rec_w<1, FIRST>(recw, t, s, g, st, chan, post, rec_in, rec_out, msg, unsat, run);
  }
#ifdef LDPC_EXPERIMENTS
  // continuous batching: the STREAM variant (never FIRST), 8 loads in flight
  static void cn_rec_stream(uint32_t vec, uint32_t recw, const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                            const T *chan, T *post, const T *rec_in, T *rec_out, T *msg, uint32_t *unsat, uint32_t run) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    auto go = [&](auto k) { k<<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, post, rec_in, rec_out, msg, unsat, run LDPC_DBG_ARG(0u)); };
    if (vec == 4 && kMaxVec == 4) {
      if (recw == 3) go(dev::cn_minsum_rec_kernel<T, kMaxVec, 3, 8, false, true, true>); else go(dev::cn_minsum_rec_kernel<T, kMaxVec, 4, 8, false, true, true>);
    } else {
      if (recw == 3) go(dev::cn_minsum_rec_kernel<T, 2, 3, 8, false, true, true>); else go(dev::cn_minsum_rec_kernel<T, 2, 4, 8, false, true, true>);
    }
  }
#endif
  static void vn_free_rec(uint32_t vec, uint32_t recw, const Tiling &t, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const uint32_t *free_rs, const T *chan, const T *rec, T *post,
                          int32_t event_iteration) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    auto go = [&](auto k) { k<<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, free_rs, chan, rec, post, event_iteration); };
    if (vec == 4 && kMaxVec == 4) {
      if (recw == 3) go(dev::vn_free_rec_kernel<T, kMaxVec, 3>); else go(dev::vn_free_rec_kernel<T, kMaxVec, 4>);
    } else if (vec >= 2) {
      if (recw == 3) go(dev::vn_free_rec_kernel<T, 2, 3>); else go(dev::vn_free_rec_kernel<T, 2, 4>);
    } else {
      if (recw == 3) go(dev::vn_free_rec_kernel<T, 1, 3>); else go(dev::vn_free_rec_kernel<T, 1, 4>);
    }
  }

  template <bool FIRST>
  static void cn_minsum(uint32_t vec, bool wide_mask, uint32_t unroll, const Tiling &t, hipStream_t s,
                        const dev::Graph &g, const dev::State &st, const T *L, T *msg, uint32_t *unsat) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      cn_minsum_m<kMaxVec, FIRST>(wide_mask, unroll, t, s, g, st, L, msg, unsat);
    else if (vec >= 2)
      cn_minsum_m<2, FIRST>(wide_mask, unroll, t, s, g, st, L, msg, unsat);
    else
      cn_minsum_m<1, FIRST>(wide_mask, unroll, t, s, g, st, L, msg, unsat);
  }

  // flooding, LDS-staged rules
  template <int RULE, bool FIRST>
  static void cn_staged_r(const Tiling &t, size_t lds, hipStream_t s, const dev::Graph &g, const dev::State &st,
                          const T *L, T *msg, uint32_t *unsat, uint32_t dmax) {
    if (g_knobs.row_scratch) {
      dev::cn_staged_kernel<RULE, T, FIRST, true><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, L, msg, unsat, dmax,
                                                                                 static_cast<T *>(g_knobs.row_scratch));
      return;
    }
    auto k = dev::cn_staged_kernel<RULE, T, FIRST>;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds));
    k<<<t.blocks, t.threads, lds, s>>>(g, t.sched, st, L, msg, unsat, dmax, nullptr);
  }
  // reg_dmax: 0 = cn_staged_kernel; 10 / 12 = cn_reg_kernel (the Tanh rule: rows of at most that many edges in registers; recs: their records)
  template <int RULE, bool FIRST>
  static void cn_staged_r(uint32_t reg_dmax, const uint32_t *recs, const Tiling &t, size_t lds, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const T *L, T *msg, uint32_t *unsat, uint32_t dmax) {
    if (reg_dmax == 0) return cn_staged_r<RULE, FIRST>(t, lds, s, g, st, L, msg, unsat, dmax);
    auto launch = [&](auto k) {
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      k<<<t.blocks, t.threads, lds, s>>>(g, t.sched, st, recs, L, msg, unsat, dmax);
    };
    if constexpr (RULE == dev::kRuleTanh || RULE == dev::kRuleTanhFast) {
      if (reg_dmax == 10)
        launch(dev::cn_reg_kernel<RULE, T, 10, FIRST>);
      else
        launch(dev::cn_reg_kernel<RULE, T, 12, FIRST>);
    }
  }
  template <bool FIRST>
  static void cn_staged(Rule rule, uint32_t reg_dmax, const uint32_t *recs, const Tiling &t, size_t lds, hipStream_t s, const dev::Graph &g,
                        const dev::State &st, const T *L, T *msg, uint32_t *unsat, uint32_t dmax) {
    switch (rule) {
      case Rule::Phi:
        if constexpr (sizeof(T) == 4) {
          if (g_knobs.fast) {
            cn_staged_r<dev::kRulePhiFast, FIRST>(reg_dmax, recs, t, lds, s, g, st, L, msg, unsat, dmax);
            break;
          }
        }
        cn_staged_r<dev::kRulePhi, FIRST>(reg_dmax, recs, t, lds, s, g, st, L, msg, unsat, dmax);
        break;
      case Rule::Tanh:
        if constexpr (sizeof(T) == 4) {
          if (g_knobs.fast) {
            cn_staged_r<dev::kRuleTanhFast, FIRST>(reg_dmax, recs, t, lds, s, g, st, L, msg, unsat, dmax);
            break;
          }
        }
        cn_staged_r<dev::kRuleTanh, FIRST>(reg_dmax, recs, t, lds, s, g, st, L, msg, unsat, dmax);
        break;
      case Rule::Minstarapprox:
        cn_staged_r<dev::kRuleMinstarapprox, FIRST>(reg_dmax, recs, t, lds, s, g, st, L, msg, unsat, dmax);
        break;
      case Rule::Aminstar:
        cn_staged_r<dev::kRuleAminstar, FIRST>(reg_dmax, recs, t, lds, s, g, st, L, msg, unsat, dmax);
        break;
      case Rule::Minsum:
        cn_staged_r<dev::kRuleMinsum, FIRST>(reg_dmax, recs, t, lds, s, g, st, L, msg, unsat, dmax);
        break;
    }
  }

  // variable nodes (list = true: only the variables of Graph::list_*)
  template <int VEC, bool LIST>
  static void vn_l(uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                   const T *chan, const T *msg, T *post, const uint32_t *unsat_in, uint32_t *unsat_clear,
                   int32_t latch_it) {
    if (g_knobs.nt_vn) {
      if (unroll >= 8)
        dev::vn_kernel<T, VEC, 8, true, LIST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post,
                                                                            unsat_in, unsat_clear, latch_it);
      else
        dev::vn_kernel<T, VEC, 4, true, LIST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post,
                                                                            unsat_in, unsat_clear, latch_it);
    } else {
      if (unroll >= 8)
        dev::vn_kernel<T, VEC, 8, false, LIST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post,
                                                                             unsat_in, unsat_clear, latch_it);
      else
        dev::vn_kernel<T, VEC, 4, false, LIST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post,
                                                                             unsat_in, unsat_clear, latch_it);
    }
  }
  template <int VEC>
  static void vn_v(bool list, uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g,
                   const dev::State &st, const T *chan, const T *msg, T *post, const uint32_t *unsat_in,
                   uint32_t *unsat_clear, int32_t latch_it) {
    if (list)
      vn_l<VEC, true>(unroll, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
    else
      vn_l<VEC, false>(unroll, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
  }
  static void vn(bool list, uint32_t vec, uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g,
                 const dev::State &st, const T *chan, const T *msg, T *post, const uint32_t *unsat_in,
                 uint32_t *unsat_clear, int32_t latch_it) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      vn_v<kMaxVec>(list, unroll, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
    else if (vec >= 2)
      vn_v<2>(list, unroll, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
    else
      vn_v<1>(list, unroll, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
  }

  // the list variant that also rebuilds the L-free posteriors of a slice's first convergences (kernels_flooding.hip.h, EVW)
  template <int VEC, int EVW>
  static void vn_event_v(uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                         const T *chan, const T *msg, T *post, const uint32_t *unsat_in, uint32_t *unsat_clear,
                         int32_t latch_it, const dev::VnEvent<T> &ev) {
    if (g_knobs.nt_vn) {
      if (unroll >= 8)
        dev::vn_kernel<T, VEC, 8, true, true, EVW><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
      else
        dev::vn_kernel<T, VEC, 4, true, true, EVW><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
    } else {
      if (unroll >= 8)
        dev::vn_kernel<T, VEC, 8, false, true, EVW><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
      else
        dev::vn_kernel<T, VEC, 4, false, true, EVW><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
    }
  }
  static void vn_event(uint32_t vec, uint32_t recw, uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g,
                       const dev::State &st, const T *chan, const T *msg, T *post, const uint32_t *unsat_in,
                       uint32_t *unsat_clear, int32_t latch_it, const dev::VnEvent<T> &ev) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    auto go = [&](auto vecc) {
      constexpr int V = decltype(vecc)::value;
      if (recw == 3)
        vn_event_v<V, 3>(unroll, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
      else
        vn_event_v<V, 4>(unroll, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
    };
    if (vec == 4 && kMaxVec == 4)
      go(std::integral_constant<int, kMaxVec>{});
    else if (vec >= 2)
      go(std::integral_constant<int, 2>{});
    else
      go(std::integral_constant<int, 1>{});
  }

  // layered
  // reg_dmax: 0 = two-pass kernel; 10 / 12 / 24 = register-resident rows of at most that many edges
  template <int RULE, bool FIRST>
  static void hl_rr(uint32_t reg_dmax, const Tiling &t, size_t lds, hipStream_t s, const dev::Graph &g,
                    const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R, uint32_t dmax) {
    auto launch = [&](auto k) {
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
      k<<<t.blocks, t.threads, lds, s>>>(g, t.sched, st, level_rows, n_level, Q, R, dmax);
    };
    if (reg_dmax == 10)
      launch(dev::hl_level_reg_kernel<RULE, T, 10, FIRST>);
    else if (reg_dmax == 12)
      launch(dev::hl_level_reg_kernel<RULE, T, 12, FIRST>);
    else if (reg_dmax == 24)
      launch(dev::hl_level_reg_kernel<RULE, T, 24, FIRST>);
    else if (g_knobs.row_scratch) {
      dev::hl_level_kernel<RULE, T, FIRST, true><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R, dmax,
                                                                                static_cast<T *>(g_knobs.row_scratch));
    } else {
      auto k = dev::hl_level_kernel<RULE, T, FIRST>;
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
      k<<<t.blocks, t.threads, lds, s>>>(g, t.sched, st, level_rows, n_level, Q, R, dmax, nullptr);
    }
  }
  template <bool FIRST>
  static void hl(Rule rule, uint32_t reg_dmax, const Tiling &t, size_t lds, hipStream_t s, const dev::Graph &g,
                 const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R, uint32_t dmax) {
    switch (rule) {
      case Rule::Phi:
        if constexpr (sizeof(T) == 4) {
          if (g_knobs.fast) {
            hl_rr<dev::kRulePhiFast, FIRST>(reg_dmax, t, lds, s, g, st, level_rows, n_level, Q, R, dmax);
            break;
          }
        }
        hl_rr<dev::kRulePhi, FIRST>(reg_dmax, t, lds, s, g, st, level_rows, n_level, Q, R, dmax);
        break;
      case Rule::Tanh:
        if constexpr (sizeof(T) == 4) {
          if (g_knobs.fast) {
            hl_rr<dev::kRuleTanhFast, FIRST>(reg_dmax, t, lds, s, g, st, level_rows, n_level, Q, R, dmax);
            break;
          }
        }
        hl_rr<dev::kRuleTanh, FIRST>(reg_dmax, t, lds, s, g, st, level_rows, n_level, Q, R, dmax);
        break;
      case Rule::Minstarapprox:
        hl_rr<dev::kRuleMinstarapprox, FIRST>(reg_dmax, t, lds, s, g, st, level_rows, n_level, Q, R, dmax);
        break;
      case Rule::Aminstar:
        hl_rr<dev::kRuleAminstar, FIRST>(reg_dmax, t, lds, s, g, st, level_rows, n_level, Q, R, dmax);
        break;
      case Rule::Minsum:
        hl_rr<dev::kRuleMinsum, FIRST>(reg_dmax, t, lds, s, g, st, level_rows, n_level, Q, R, dmax);
        break;
    }
  }

  // layered, slice-persistent (hl_slice_kernel): one launch per iteration; f32 Tanh rule (and its "@fast" variant)
  struct SliceLaunch {
    uint32_t slice, blocks, columns, dmax, n_levels, tile;
    size_t lds;
    const uint32_t *tasks, *task_ptr;
  };
  static constexpr uint32_t kSliceThreads = 1024;
#ifdef LDPC_EXPERIMENTS
  template <int RULE, bool FIRST>
  static void hl_slice_r(const SliceLaunch &p, hipStream_t s, const dev::Graph &g, const dev::State &st, T *Q, T *R) {
    if constexpr (sizeof(T) == 4) {
      auto launch = [&](auto k) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(p.lds));
        k<<<p.blocks, kSliceThreads, p.lds, s>>>(g, st, p.tasks, p.task_ptr, p.n_levels, p.tile, Q, R, p.dmax, p.columns);
      };
      if (p.slice == 32)
        launch(dev::hl_slice_kernel<RULE, T, 32, kSliceThreads, FIRST>);
      else
        launch(dev::hl_slice_kernel<RULE, T, 64, kSliceThreads, FIRST>);
    }
  }
  template <bool FIRST>
  static void hl_slice(const SliceLaunch &p, hipStream_t s, const dev::Graph &g, const dev::State &st, T *Q, T *R) {
    if (g_knobs.fast)
      hl_slice_r<dev::kRuleTanhFast, FIRST>(p, s, g, st, Q, R);
    else
      hl_slice_r<dev::kRuleTanh, FIRST>(p, s, g, st, Q, R);
  }
#endif

  // layered min-sum, streaming
  template <int VEC, bool FIRST>
  static void hl_minsum_v(uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R) {
    if (unroll >= 8)
      dev::hl_minsum_kernel<T, VEC, 8, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R);
    else
      dev::hl_minsum_kernel<T, VEC, 4, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R);
  }
  // register-resident rows: DMAX bucket of the level's largest row; vec capped so that the
  // 2 * DMAX * VEC values fit the register file with some occupancy left
  // 0 = no register-resident form for this level (rows too long for the register budget even with
  // one codeword per lane: the two-pass kernel takes it)
  static uint32_t hl_reg_bucket(uint32_t maxdeg) {
    const uint32_t dmax = maxdeg <= 8 ? 8 : (maxdeg <= 12 ? 12 : (maxdeg <= 20 ? 20 : (maxdeg <= 32 ? 32 : 0)));
    return 2 * dmax * (sizeof(T) / 4) <= 96 ? dmax : 0;
  }
  static uint32_t hl_reg_vec(uint32_t vec, uint32_t dmax) {
    const uint32_t words = sizeof(T) / 4;
    while (vec > 1 && 2 * dmax * vec * words > 96) vec /= 2;
    return vec;
  }
  template <int VEC, bool FIRST>
  // returns false when the (VEC, DMAX) pair has no instantiation (the caller must not let that pass)
  static bool hl_minsum_reg_v(uint32_t dmax, const Tiling &t, hipStream_t s, const dev::Graph &g,
                              const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R) {
    switch (dmax) {
      case 8:
        dev::hl_minsum_reg_kernel<T, VEC, 8, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R);
        return true;
      case 12:
        dev::hl_minsum_reg_kernel<T, VEC, 12, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R);
        return true;
      case 20:
        if constexpr (VEC * sizeof(T) <= 8) {
          dev::hl_minsum_reg_kernel<T, VEC, 20, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R);
          return true;
        }
        return false;
      case 32:
        if constexpr (VEC * sizeof(T) <= 4) {
          dev::hl_minsum_reg_kernel<T, VEC, 32, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R);
          return true;
        }
        return false;
      default:
        return false;
    }
  }
  template <bool FIRST>
  static bool hl_minsum_reg(uint32_t vec, uint32_t dmax, const Tiling &t, hipStream_t s, const dev::Graph &g,
                            const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4) return hl_minsum_reg_v<kMaxVec, FIRST>(dmax, t, s, g, st, level_rows, n_level, Q, R);
    if (vec >= 2) return hl_minsum_reg_v<2, FIRST>(dmax, t, s, g, st, level_rows, n_level, Q, R);
    return hl_minsum_reg_v<1, FIRST>(dmax, t, s, g, st, level_rows, n_level, Q, R);
  }
  // layered min-sum with row records (hl_minsum_rec_kernel; three-word records only): the row's Qv values and two
  // records live in registers
  static uint32_t hl_rec_vec(uint32_t vec, uint32_t dmax) {
    const uint32_t words = sizeof(T) / 4;
    while (vec > 1 && (dmax + 6) * vec * words > 112) vec /= 2;
    return vec;
  }
  template <int VEC, bool FIRST>
  static bool hl_minsum_rec_v(uint32_t dmax, const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                              const uint32_t *level_rows, uint32_t n_level, T *Q, T *rec) {
    constexpr uint32_t kWords = VEC * sizeof(T) / 4;
    switch (dmax) {
      case 8:
        dev::hl_minsum_rec_kernel<T, VEC, 8, 3, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, rec);
        return true;
      case 12:
        dev::hl_minsum_rec_kernel<T, VEC, 12, 3, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, rec);
        return true;
      case 20:
        if constexpr ((20 + 6) * kWords <= 112) {
          dev::hl_minsum_rec_kernel<T, VEC, 20, 3, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, rec);
          return true;
        }
        return false;
      case 32:
        if constexpr ((32 + 6) * kWords <= 112) {
          dev::hl_minsum_rec_kernel<T, VEC, 32, 3, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, rec);
          return true;
        }
        return false;
      default:
        return false;
    }
  }
  template <bool FIRST>
  static bool hl_minsum_rec(uint32_t vec, uint32_t dmax, const Tiling &t, hipStream_t s, const dev::Graph &g,
                            const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *rec) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4) return hl_minsum_rec_v<kMaxVec, FIRST>(dmax, t, s, g, st, level_rows, n_level, Q, rec);
    if (vec >= 2) return hl_minsum_rec_v<2, FIRST>(dmax, t, s, g, st, level_rows, n_level, Q, rec);
    return hl_minsum_rec_v<1, FIRST>(dmax, t, s, g, st, level_rows, n_level, Q, rec);
  }
  template <bool FIRST>
  static void hl_minsum(uint32_t vec, uint32_t unroll, const Tiling &t, hipStream_t s, const dev::Graph &g,
                        const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      hl_minsum_v<kMaxVec, FIRST>(unroll, t, s, g, st, level_rows, n_level, Q, R);
    else if (vec >= 2)
      hl_minsum_v<2, FIRST>(unroll, t, s, g, st, level_rows, n_level, Q, R);
    else
      hl_minsum_v<1, FIRST>(unroll, t, s, g, st, level_rows, n_level, Q, R);
  }
};

// LDS-staged kernels: largest block whose [arrays][dmax][threads] columns fit the CU's LDS
bool staged_block(uint32_t arrays, uint32_t dmax, size_t elem, uint32_t *threads, size_t *lds) {
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
size_t scratch_bytes_for(const Tiling &t, uint32_t dmax, size_t elem) {
  return size_t(t.blocks) * (t.threads / 64) * 2 * dmax * 64 * elem;
}

}  // namespace

namespace {

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

}  // namespace

// Codewords per lane (1, 2 or 4) of the streaming kernels: a wave covers 64 * vec codewords, and
// those slices must tile the layout tile exactly (a 192-codeword tile takes vec = 1: with 128-wide
// slices its last 64 codewords would belong to no wave).
static uint32_t pick_vec_for(uint32_t tile, uint32_t max_vec, uint32_t wanted) {
  uint32_t vec = std::min<uint32_t>(std::min(max_vec, std::max<uint32_t>(wanted, 1)), 4);
  if (vec == 3) vec = 2;
  while (vec > 1 && tile % (64 * vec) != 0) vec /= 2;
  return vec;
}

// ---- one group of codewords ----------------------------------------------------------------

template <typename T>
int DeviceDecoder::run_group(Workspace &w, const void *llrs, bool llrs_f64, size_t nb, uint32_t max_iterations,
                             uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior,
                             hipStream_t s, bool may_block) {
  const uint32_t G = static_cast<uint32_t>(w.G);
  const uint32_t W = G / 64;
  const uint32_t n = static_cast<uint32_t>(n_), m = static_cast<uint32_t>(m_);
  T *chan = static_cast<T *>(w.chan), *post = static_cast<T *>(w.post), *msg = static_cast<T *>(w.msg);
  // default: enough waves that each handles ~4 nodes (oversubscription evens out the tail)
  const uint32_t target_waves = opt_waves_ ? opt_waves_ : 256 * 1024;
  // LDS columns per thread of the staged kernels: the Tanh rule works in one (rule_check_node), the others need
  // the inputs beside the outputs
  const uint32_t lds_columns = impl_.rule == Rule::Tanh ? 1u : 2u;
  const uint32_t unroll = opt_unroll_cn_;
  const uint32_t unroll_vn = opt_unroll_vn_;

  // layout tile: codewords per self-contained sub-batch (kernels.hip.h, tile_base)
  uint32_t tile = opt_tile_ ? opt_tile_ : (sizeof(T) == 4 ? 256 : 128);
  tile = std::max<uint32_t>(64, tile / 64 * 64);
  while (G % tile != 0) tile -= 64;

  g_knobs.lfree_unroll = opt_lfree_unroll_;
  g_knobs.rec_unroll = opt_rec_unroll_;
  g_knobs.rec_dbg = opt_rec_dbg_;
  g_knobs.rec_long = max_row_weight_ > 8 || opt_rec_long_;
  g_knobs.fast = impl_.fast;
  g_knobs.lfree_nt_in = opt_lfree_nt_in_;
  g_knobs.nt = opt_nt_;
  g_knobs.nt_vn = opt_nt_vn_;
  g_knobs.row_scratch = nullptr;
  dev::Graph g{d_row_ptr_, d_edge_col_, d_col_ptr_, d_col_edge_, m, n, static_cast<uint32_t>(e_),
               nullptr,    nullptr,     nullptr,    0,           d_edge_aux_, d_edge_peer_};
  dev::State st{w.done, w.iters, w.n_active, w.n_slots, w.slot_cw, nullptr, 0, 0, nullptr, nullptr, 0};
  // progress word: the first check-node launch of iteration `it` runs with ticked(it)
  w.epoch = (w.epoch % 0xFFFFFFu) + 1;
  auto ticked = [&](uint32_t it) {
    dev::State t = st;
    t.publish = opt_poll_ ? w.d_flag : nullptr;
    t.epoch = w.epoch;
    t.tick = it;
    return t;
  };
  const ProgressPoll poll{(opt_poll_ && w.d_flag) ? w.h_flag : nullptr, w.epoch, may_block,
                          t_pace_lead ? t_pace_lead : (impl_.schedule == Schedule::Layered ? 2u : 8u), s};

  dev::init_group_kernel<<<(G + 255) / 256, 256, 0, s>>>(w.done, w.iters, w.unsat0, w.unsat1, w.n_active, w.n_slots,
                                                         w.slot_cw, static_cast<uint32_t>(nb), G);
  {
    dim3 grid((n + 63) / 64, W);
    const uint32_t block_size = pattern_len_ ? n / pattern_len_ : 0;
    if (llrs_f64)
      dev::ingest_kernel<double, T><<<grid, 256, 0, s>>>(static_cast<const double *>(llrs), input_len_,
                                                        static_cast<uint32_t>(nb), n, G, tile, chan, post,
                                                        w.rawbits, d_src_block_, block_size);
    else
      dev::ingest_kernel<float, T><<<grid, 256, 0, s>>>(static_cast<const float *>(llrs), input_len_,
                                                       static_cast<uint32_t>(nb), n, G, tile, chan, post,
                                                       w.rawbits, d_src_block_, block_size);
    if (w.after_ingest) {
      HIP_TRY(hipEventRecord(w.after_ingest, s));
      if (w.ingest_seq) w.ingest_seq->fetch_add(1, std::memory_order_release);
    }
  }
  // enough threads to fill the chip: each handles one packed word of a few checks
  // a wavefront takes 64 packed words of a few checks; enough wavefronts to fill the chip
  const uint32_t synd_chunks = (W + 63) / 64;
  const uint32_t synd_rows =
      std::max<uint32_t>(1, std::min<uint32_t>(64, uint32_t(uint64_t(m) * synd_chunks * 64 / opt_synd_threads_)));
  const uint32_t synd_threads = 64 * synd_chunks * ((m + synd_rows - 1) / synd_rows);
  auto syndrome_of = [&](const uint64_t *hard, uint32_t *unsat) {
    if (m == 0) return;
    dev::syndrome_bits_kernel<<<(synd_threads + 255) / 256, 256, 0, s>>>(d_row_ptr_, d_edge_col_, m, hard,
                                                                         unsat, w.n_active, w.n_slots, W, synd_rows);
  };
  auto latch = [&](uint32_t *unsat, int32_t it) {
    dev::latch_kernel<<<(G + 255) / 256, 256, 0, s>>>(w.done, w.iters, unsat, w.n_active, it, G);
  };
  // (one codeword per lane: the paired-load form of the 16-bit posterior, pack_hard_pair_kernel, is slower here --
  // 210 vs 191 us for 8192 x BG1 Zc=384: 256-byte requests already stream, the exchange only adds work)
  // (16 K waves by default: the launch runs once per layered iteration and its waves are short -- at the 256 K of the other
  // launches a wave packs six rows and is gone; 5G NR BG1 Zc=384 HLTanhf32 +0.7 % over three alternating pairs, round 5)
  const Tiling pack_t = make_tiling(G, tile, 64, n, 256, opt_waves_pack_ ? opt_waves_pack_ : std::min<uint32_t>(target_waves, 16384));
  auto pack = [&](const T *soft) {
    dev::pack_hard_kernel<T><<<pack_t.blocks, pack_t.threads, 0, s>>>(soft, w.hardbits, w.n_active, w.n_slots, n, tile,
                                                                      W, pack_t.sched.waves_per_chunk);
  };

  auto emit = [&](int zero_fill, int retire_only) {
    dim3 grid(std::min<uint32_t>((n + 63) / 64, retire_only ? opt_retire_blocks_ : 4096), W);
    if (llrs_f64)
      dev::emit_kernel<T, double><<<grid, 256, 0, s>>>(post, w.rawbits, st, &w.plan->do_compact, n, G, tile,
                                                      static_cast<uint32_t>(out_len), bits, iterations,
                                                      static_cast<double *>(posterior), zero_fill, retire_only);
    else
      dev::emit_kernel<T, float><<<grid, 256, 0, s>>>(post, w.rawbits, st, &w.plan->do_compact, n, G, tile,
                                                     static_cast<uint32_t>(out_len), bits, iterations,
                                                     static_cast<float *>(posterior), zero_fill, retire_only);
  };
  // batch compaction checkpoint (kernels.hip.h): everything decided on the device
  const Tiling mv_t = make_tiling(G, tile, 64, n, 256, opt_move_waves_);
  uint32_t post_move_rows = 0;  // 0 = all rows; set by the flooding L-free paths below
  auto compact = [&](uint32_t remaining, T *msg_cur, bool with_chan, uint32_t msg_rows) {
    dev::compact_plan_kernel<<<1, 1024, 0, s>>>(
        ticked(max_iterations - remaining), w.plan, w.perm, w.slot_tmp, w.fill_cw, remaining,
        dev::CompactRule{opt_compact_horizon_, opt_compact_cost_live_, opt_compact_cost_slots_, opt_compact_min_freed_q_});
    emit(0, 1);
    dev::MoveList<T> ml{};
    auto add = [&](T *arr, uint32_t rows, uint32_t moved) {
      ml.arr[ml.count] = arr;
      ml.rows[ml.count] = rows;
      ml.moved[ml.count++] = moved;
    };
    if (with_chan) add(chan, n, n);
    // (flooding min-sum with L-free variables: the posterior of a degree <= 2 variable is rebuilt by the next check-node pass
    // from the channel LLR and the records / messages before anything reads it -- every slice stores them after a commit --
    // so the rows beyond the last variable the variable-node kernel writes need not travel: DVB-S2's staircase, 5G NR's
    // extension parity: half of the posterior rows, 14 % of what a mover carries)
    add(post, n, post_move_rows ? post_move_rows : n);
    if (msg_rows) add(msg_cur, msg_rows, msg_rows);
    dev::compact_move_kernel<T><<<mv_t.blocks, mv_t.threads, 0, s>>>(w.plan, w.perm, w.slot_tmp, ml, tile,
                                                                     mv_t.sched.nchunks, mv_t.sched.waves_per_chunk);
    dev::compact_commit_kernel<<<(G + 255) / 256, 256, 0, s>>>(st, w.plan, w.unsat0, w.unsat1, w.n_slots, w.fill_cw, G);
  };
  auto checkpoint_due = [&](uint32_t it) {
    if (!opt_compact_ || max_iterations < 12 || it + 4 > max_iterations) return false;
    // (0 = by schedule.  The layered schedule converges in a third of the iterations flooding needs, and a frame is
    // latched in the iteration it converges in: checkpoints from iteration 3 on, every iteration -- BASELINE config 3 at
    // +2 dB, where no frame runs more than 6 iterations: 323 k -> 331 k codewords/s, tools/c3_p2_sweep.py)
    const bool layered = impl_.schedule == Schedule::Layered;
    const uint32_t first = opt_compact_first_ ? opt_compact_first_ : (layered ? 3u : 6u);
    const uint32_t every = opt_compact_every_ ? opt_compact_every_ : (layered ? 1u : 2u);
    if (it < first) return false;
    return it <= 26 ? (it - first) % every == 0 : it % 4 == 0;
  };

  // pre-check on the raw input: iterations = 0 (flooding.rs:57-64)
  syndrome_of(w.rawbits, w.unsat0);
  latch(w.unsat0, 0);

  uint32_t *unsat[2] = {w.unsat0, w.unsat1};
  int zero_fill = 0;

  if (impl_.schedule == Schedule::Flooding) {
    // the streaming min-sum kernels keep a row's signs in a 64-bit mask: longer rows take the
    // LDS-staged kernel
    const bool streaming = impl_.rule == Rule::Minsum && !opt_staged_minsum_ && max_row_weight_ <= 64;
    const uint32_t vec = pick_vec_for(tile, sizeof(T) == 4 ? 4 : 2, opt_vec_);
    uint32_t stream_block = opt_block_;
    if (stream_block != 64 && stream_block != 128) stream_block = 256;
    const Tiling vn_t = make_tiling(G, tile, 64 * vec, n, stream_block, opt_waves_vn_ ? opt_waves_vn_ : (opt_waves_ ? opt_waves_ : 128 * 1024));
    Tiling cn_t = make_tiling(G, tile, 64 * vec, m, stream_block, target_waves);
    uint32_t st_threads = 256;
    size_t st_lds = 0;
    if (!streaming) {
      if (staged_block(lds_columns, max_row_weight_, sizeof(T), &st_threads, &st_lds)) {
        cn_t = make_tiling(G, tile, 64, m, st_threads, target_waves);
      } else {
        // rows beyond the LDS: the columns live in HBM, one region per wavefront of a small launch
        st_threads = kScratchThreads;
        st_lds = 0;
        cn_t = make_tiling(G, tile, 64, m, st_threads, std::min(target_waves, kScratchWaves));
        if (int rc = ensure_row_scratch(w, scratch_bytes_for(cn_t, max_row_weight_, sizeof(T)))) return rc;
        g_knobs.row_scratch = w.row_scratch;
      }
    }
    // the Tanh rule on graphs with rows of at most 12 edges: rows in registers (cn_reg_kernel: 32-bit byte offsets inside
    // a tile slice).  Measured (round 4, 0.xxx of the roofline, cn_staged_kernel -> cn_reg_kernel): DVB-S2 1/2 Tanhf32
    // 0.455 -> 0.469, Tanhf64 0.410 -> 0.417, CCSDS AR4JA 1/2 Tanhf32 0.478 -> 0.479; the other rules lose 0-2 % and 5G NR
    // BG1's mixed 3..19-edge rows in one 24-edge bucket 15 %, so they keep cn_staged_kernel.
    const uint32_t cn_reg = (streaming || !opt_cn_reg_ || d_row_recs_ == nullptr || impl_.rule != Rule::Tanh ||
                             uint64_t(std::max(e_, n_)) * tile * sizeof(T) >= (1ull << 32))
                                ? 0u
                                : (max_row_weight_ <= 10 ? 10u : (max_row_weight_ <= 12 ? 12u : 0u));
    const bool wide_mask = max_row_weight_ > 32;
    // row records instead of per-edge messages on the check-node side (kernels.hip.h, cn_minsum_rec_kernel)
    // (its buffer addressing carries 32-bit byte offsets inside a tile slice)
    const bool records = streaming && w.records && w.rec[0] != nullptr &&
                         uint64_t(std::max<size_t>(std::max(e_, n_), m_ * rec_w_)) * tile * sizeof(T) < (1ull << 32);
    const bool lfree = streaming && lfree_ready_ && opt_lfree_ && (w.msg2 != nullptr || records);
    T *mbuf[2] = {msg, (lfree && !records) ? static_cast<T *>(w.msg2) : msg};
    T *rbuf[2] = {static_cast<T *>(w.rec[0]), static_cast<T *>(w.rec[1])};
    if (lfree && post_rows_keep_ > 0 && post_rows_keep_ <= n) post_move_rows = post_rows_keep_;
    const bool quiet = records && opt_rec_quiet_;
    if (quiet) {
      st.slice_state = w.slice_state;
      HIP_TRY(hipMemsetAsync(w.slice_state, 0, size_t(G / 64) * sizeof(uint32_t), s));
    }
    const uint32_t rec_run = std::max<uint32_t>(1, std::min<uint32_t>(opt_rec_run_, m));
    const Tiling rec_t = make_tiling(G, tile, 64 * vec, (m + rec_run - 1) / rec_run, stream_block, target_waves);
    dev::Graph g_keep = g, g_free = g;
    Tiling vn_keep_t = vn_t, vn_free_t = vn_t, vn_event_t = vn_t;
    if (lfree) {
      g_keep.list_var = d_keep_var_;
      g_keep.list_ptr = d_keep_ptr_;
      g_keep.list_edge = records ? d_keep_pos_ : d_keep_edge_;  // records: the messages are stored in this list's order
      g_keep.n_list = n_keep_;
      g_free.list_var = d_free_var_;
      g_free.list_ptr = d_free_ptr_;
      g_free.list_edge = d_free_edge_;
      g_free.n_list = n_free_;
      const uint32_t wv = opt_waves_vn_ ? opt_waves_vn_ : (opt_waves_ ? opt_waves_ : 128 * 1024);
      vn_keep_t = make_tiling(G, tile, 64 * vec, n_keep_, stream_block, wv);
      vn_keep_t.sched.reverse = opt_vn_reverse_ ? 1u : 0u;
      vn_free_t = make_tiling(G, tile, 64 * vec, n_free_, stream_block, wv);
      vn_event_t = make_tiling(G, tile, 64 * vec, n_free_, stream_block, 16 * 1024);
    }
    // a paced host (run_any) sees the group's running count: once the first codewords have converged, every iteration
    // ends with a checkpoint (the device still decides whether re-packing pays)
    const bool adaptive = opt_compact_ && poll.flag != nullptr && poll.throttle && !opt_compact_every_;
    bool seen_drop = false;
    auto tail_checkpoint = [&](uint32_t it) {
      const uint32_t first_ck = opt_compact_first_ ? opt_compact_first_ : 6u;
      return seen_drop && max_iterations >= 12 && it + 4 <= max_iterations && it >= first_ck;
    };
    for (uint32_t it = 1; it <= max_iterations; it++) {
      if (it > 1 && poll.finished(it)) break;  // everything below would return at once
      if (adaptive && !seen_drop && poll.running(static_cast<uint32_t>(nb)) < nb) seen_drop = true;
      const bool first = it == 1;
      uint32_t *unsat_out = unsat[it & 1];
      T *m_out = mbuf[it & 1];
      const T *m_in = mbuf[(it + 1) & 1];
      const dev::State stp = ticked(it);
      timed_begin(kKernelCheck, s);
      if (records) {
        if (first)
          Launch<T>::template cn_rec<true>(vec, rec_w_, rec_t, s, g, stp, chan, post, rbuf[(it + 1) & 1], rbuf[it & 1], msg,
                                           unsat_out, rec_run);
        else
          Launch<T>::template cn_rec<false>(vec, rec_w_, rec_t, s, g, stp, chan, post, rbuf[(it + 1) & 1], rbuf[it & 1], msg,
                                            unsat_out, rec_run);
      } else if (lfree) {
        if (first)
          Launch<T>::template cn_lfree<true>(vec, wide_mask, cn_t, s, g, stp, chan, post, m_in, m_out, unsat_out);
        else
          Launch<T>::template cn_lfree<false>(vec, wide_mask, cn_t, s, g, stp, chan, post, m_in, m_out, unsat_out);
      } else if (streaming) {
        if (first)
          Launch<T>::template cn_minsum<true>(vec, wide_mask, unroll, cn_t, s, g, stp, chan, msg, unsat_out);
        else
          Launch<T>::template cn_minsum<false>(vec, wide_mask, unroll, cn_t, s, g, stp, post, msg, unsat_out);
      } else {
        if (first)
          Launch<T>::template cn_staged<true>(impl_.rule, cn_reg, d_row_recs_, cn_t, st_lds, s, g, stp, chan, msg, unsat_out,
                                              max_row_weight_);
        else
          Launch<T>::template cn_staged<false>(impl_.rule, cn_reg, d_row_recs_, cn_t, st_lds, s, g, stp, post, msg, unsat_out,
                                               max_row_weight_);
      }
      timed_end(kKernelCheck, s);
      if (first && skew_record_) {
        HIP_TRY(hipEventRecord(skew_record_, s));
        skew_record_ = nullptr;
      }
      timed_begin(kKernelVar, s);
      // (deferred L-free stores: the first convergences of a slice get their L-free posteriors from the records of the latched
      // iteration INSIDE this launch -- rounds 3-4 ran a small vn_free_rec_kernel launch behind it in every iteration, which
      // almost always found nothing: 4.4 us + a 5.7 us dispatch gap per iteration)
      if (quiet && it > 1 && opt_vn_event_) {
        const dev::VnEvent<T> ev{d_free_var_, d_free_rs_, rbuf[(it - 1) & 1], n_free_};
        Launch<T>::vn_event(vec, rec_w_, unroll_vn, vn_keep_t, s, g_keep, st, chan, m_out, post, unsat_out, unsat[(it + 1) & 1],
                            static_cast<int32_t>(it) - 1, ev);
      } else {
        Launch<T>::vn(lfree, vec, unroll_vn, lfree ? vn_keep_t : vn_t, s, lfree ? g_keep : g, st, chan, m_out, post,
                      first ? nullptr : unsat_out, unsat[(it + 1) & 1], static_cast<int32_t>(it) - 1);
        if (quiet && it > 1)
          Launch<T>::vn_free_rec(vec, rec_w_, vn_event_t, s, g_free, st, d_free_rs_, chan, rbuf[(it - 1) & 1], post,
                                 static_cast<int32_t>(it) - 1);
      }
      timed_end(kKernelVar, s);
      if (checkpoint_due(it) || tail_checkpoint(it)) {
        // what the next iteration reads: the records of this one (the per-edge messages have been consumed)
        if (records)
          compact(max_iterations - it, rbuf[it & 1], true, m * rec_w_);
        else
          compact(max_iterations - it, m_out, true, static_cast<uint32_t>(e_));
      }
    }
    if (records && max_iterations > 0) {
      Launch<T>::vn_free_rec(vec, rec_w_, vn_free_t, s, g_free, st, d_free_rs_, chan, rbuf[max_iterations & 1], post, -1);
    } else if (lfree && max_iterations > 0) {
      // posterior of the L-free variables after the last iteration (no later check-node pass
      // rebuilds it): one variable-node pass over just them; frozen codewords are skipped
      dev::State st_nolatch = st;
      Launch<T>::vn(true, vec, unroll_vn, vn_free_t, s, g_free, st_nolatch, chan, mbuf[max_iterations & 1], post,
                    nullptr, w.scratch_flags, -1);
    }
    if (max_iterations > 0) {
      // syndrome of the last posterior (flooding.rs:69-79 at iteration == max_iterations)
      pack(post);
      uint32_t *u = unsat[(max_iterations + 1) & 1];
      syndrome_of(w.hardbits, u);
      latch(u, static_cast<int32_t>(max_iterations));
    } else {
      zero_fill = 1;
    }
  } else {
    uint32_t threads = 64;
    size_t lds = 0;
    if (!staged_block(lds_columns, max_row_weight_, sizeof(T), &threads, &lds) && impl_.rule != Rule::Minsum) {
      // some level has rows beyond the LDS: those levels keep their columns in HBM (a small launch, one region per wave)
      const size_t waves_bound = size_t(kScratchWaves) + size_t(G / 64) * (kScratchThreads / 64);
      if (int rc = ensure_row_scratch(w, waves_bound * 2 * max_row_weight_ * 64 * sizeof(T))) return rc;
    }
    const uint32_t n_levels = level_ptr_.empty() ? 0 : static_cast<uint32_t>(level_ptr_.size() - 1);
    const dev::State st0 = st;
    const bool streaming = impl_.rule == Rule::Minsum && !opt_staged_minsum_;
    const uint32_t vec = pick_vec_for(tile, sizeof(T) == 4 ? 4 : 2, opt_vec_);
    // Row-serial mode: when the dependency levels are (almost) single rows -- DVB-S2's staircase
    // chains every row to the next -- one launch per level is launch-bound (32 400 launches per
    // iteration).  Codewords are independent, so ONE wave per codeword slice can walk all rows in
    // level order by itself: one launch per iteration, no inter-wave ordering needed.
    const bool serial = n_levels > opt_serial_levels_;
    const uint32_t n_launch = serial ? std::min<uint32_t>(n_levels, 1) : n_levels;
    // layered min-sum: row records instead of per-edge R (kernels.hip.h, hl_minsum_rec_kernel) when every row fits the
    // three-word record and the register-resident form, and the records fit the message array
    const bool hl_rec = streaming && opt_hl_records_ && opt_hl_reg_ && max_row_weight_ <= (sizeof(T) == 4 ? 26u : 58u) &&
                        Launch<T>::hl_reg_bucket(max_row_weight_) != 0 && m_ * 3 <= e_ &&
                        uint64_t(std::max(n_, m_ * 3)) * tile * sizeof(T) < (1ull << 32);
    // slice-persistent form (kernels.hip.h, hl_slice_kernel): one launch per iteration, a workgroup per codeword slice.
    // Opt-in ("hl_persist"); for the f32 Tanh rule when every row fits a task and the slice's arrays stay below the
    // kernel's out-of-range marks (2^31 bytes; a padding index times a row's bytes must not wrap: rows of at most 1 KiB).
    typename Launch<T>::SliceLaunch sl{};
#ifdef LDPC_EXPERIMENTS
    if (sizeof(T) == 4 && impl_.rule == Rule::Tanh && opt_hl_persist_ && !serial && d_slice_tasks_[0] && tile % 64 == 0 &&
        (tile & (tile - 1)) == 0 &&  // (a padding index times a 768-byte row would wrap INTO the arrays: power-of-two rows only)
        tile * sizeof(T) <= 1024 && uint64_t(std::max(e_, n_)) * tile * sizeof(T) < (1ull << 31) && n_ < 0x003FFFFFu) {
      // (slices of 64 codewords -- a whole wavefront per row -- when that still gives every CU a workgroup and no row
      // needs splitting; else slices of 32)
      const uint32_t width = opt_hl_slice_ ? opt_hl_slice_ : ((G / 64 >= 256 && slice_fits_[1]) ? 64u : 32u);
      const int k = width == 32 ? 0 : 1;
      const uint32_t dmax_lds = std::max<uint32_t>(max_row_weight_, 10);
      const size_t lds_bytes = (size_t(dmax_lds) * sizeof(T) + 2 * 10 * 4) * Launch<T>::kSliceThreads + 16;
      if (slice_fits_[k] && lds_bytes <= size_t(160) * 1024) {
        sl.slice = width;
        sl.blocks = G / width;
        sl.columns = 1;
        sl.dmax = dmax_lds;
        sl.n_levels = n_levels;
        sl.tile = tile;
        sl.lds = lds_bytes;
        sl.tasks = d_slice_tasks_[k];
        sl.task_ptr = d_slice_task_ptr_[k];
      }
    }
#endif
    __atomic_store_n(&last_persist_, sl.slice, __ATOMIC_RELAXED);  // (both lanes' enqueuing threads pass here)
    for (uint32_t it = 1; it <= max_iterations; it++) {
      if (it > 1 && poll.finished(it)) break;
      const dev::State stp = ticked(it);
#ifdef LDPC_EXPERIMENTS
      if (sl.slice) {
        timed_begin(kKernelLayer, s);
        if (it == 1)
          Launch<T>::template hl_slice<true>(sl, s, g, stp, post, msg);
        else
          Launch<T>::template hl_slice<false>(sl, s, g, stp, post, msg);
        timed_end(kKernelLayer, s);
      }
#endif
      for (uint32_t l = 0; l < (sl.slice ? 0u : n_launch); l++) {
        const dev::State &st = l == 0 ? stp : st0;
        const uint32_t r0 = serial ? 0 : level_ptr_[l], cnt = serial ? m : level_ptr_[l + 1] - level_ptr_[l];
        const uint32_t lmaxdeg = serial ? max_row_weight_ : level_maxdeg_[l];
        const uint32_t tnodes = serial ? 1 : cnt;        // serial: one wave per slice (make_tiling: wpc = 1)
        const uint32_t sblock = serial ? 64 : 256;
        const uint32_t reg_dmax = opt_hl_reg_ ? Launch<T>::hl_reg_bucket(lmaxdeg) : 0;
        if (hl_rec) {
          // the row's messages as one record in the message array (every level qualifies, or none does)
          const uint32_t rvec = Launch<T>::hl_rec_vec(vec, reg_dmax);
          const Tiling t = make_tiling(G, tile, 64 * rvec, tnodes, sblock, target_waves);
          timed_begin(kKernelLayer, s);
          const bool launched =
              it == 1 ? Launch<T>::template hl_minsum_rec<true>(rvec, reg_dmax, t, s, g, st, d_level_rows_ + r0, cnt, post, msg)
                      : Launch<T>::template hl_minsum_rec<false>(rvec, reg_dmax, t, s, g, st, d_level_rows_ + r0, cnt, post, msg);
          timed_end(kKernelLayer, s);
          if (!launched) {
            fail("internal error: no row-record layered kernel for this level");
            return -3;
          }
          continue;
        }
        if (streaming && reg_dmax) {
          const uint32_t rvec = Launch<T>::hl_reg_vec(vec, reg_dmax);
          const Tiling t = make_tiling(G, tile, 64 * rvec, tnodes, sblock, target_waves);
          timed_begin(kKernelLayer, s);
          const bool launched =
              it == 1 ? Launch<T>::template hl_minsum_reg<true>(rvec, reg_dmax, t, s, g, st, d_level_rows_ + r0, cnt, post, msg)
                      : Launch<T>::template hl_minsum_reg<false>(rvec, reg_dmax, t, s, g, st, d_level_rows_ + r0, cnt, post, msg);
          timed_end(kKernelLayer, s);
          if (!launched) {
            fail("internal error: no register-resident layered kernel for this level");
            return -3;
          }
          continue;
        }
        if (streaming) {
          const Tiling t = make_tiling(G, tile, 64 * vec, tnodes, sblock, target_waves);
          timed_begin(kKernelLayer, s);
          if (it == 1)
            Launch<T>::template hl_minsum<true>(vec, unroll, t, s, g, st, d_level_rows_ + r0, cnt, post, msg);
          else
            Launch<T>::template hl_minsum<false>(vec, unroll, t, s, g, st, d_level_rows_ + r0, cnt, post, msg);
          timed_end(kKernelLayer, s);
          continue;
        }
        // per level: LDS columns only as tall as this level's longest row (more workgroups per
        // CU for the short-row levels), and the register-resident form when the rows fit it
        const uint32_t ldmax = std::max<uint32_t>(lmaxdeg, 1);
        uint32_t lthreads = threads;
        size_t llds = lds;
        const bool lfits = staged_block(lds_columns, ldmax, sizeof(T), &lthreads, &llds);
        if (serial) {
          lthreads = 64;
          llds = size_t(lds_columns) * ldmax * 64 * sizeof(T);
        }
        if (!lfits) {
          lthreads = serial ? 64 : kScratchThreads;
          llds = 0;
        }
        // (the register-resident form addresses Qv and R through buffer descriptors with 32-bit byte offsets
        // inside a tile slice: graphs too large for that take the two-pass kernel)
        const bool fits32 = uint64_t(std::max(e_, n_)) * tile * sizeof(T) < (1ull << 32);
        // (a 10-edge bucket beside 12 and 24: 5G NR's extension rows have at most 10 edges, and the two registers per
        // edge it saves decide whether the Tanh rule's kernel keeps 7 or 8 waves per SIMD)
        const uint32_t lreg = (!opt_hl_reg_ || !fits32) ? 0 : (ldmax <= 10 ? 10 : (ldmax <= 12 ? 12 : (ldmax <= 24 ? 24 : 0)));
        // (the register-resident kernels read a level's row records, the two-pass kernel the row list)
        const uint32_t *ltab = !lreg ? d_level_rows_ + r0 : (serial ? d_serial_recs_ : d_level_recs_ + level_rec_ptr_[l]);
        const Tiling t = make_tiling(G, tile, 64, tnodes, lthreads, lfits ? target_waves : std::min(target_waves, kScratchWaves));
        g_knobs.row_scratch = nullptr;
        if (!lfits) {
          if (scratch_bytes_for(t, ldmax, sizeof(T)) > w.row_scratch_bytes) {
            fail("internal error: row scratch smaller than a level's launch");
            return -3;
          }
          g_knobs.row_scratch = w.row_scratch;
        }
        timed_begin(kKernelLayer, s);
        if (it == 1)
          Launch<T>::template hl<true>(impl_.rule, lreg, t, llds, s, g, st, ltab, cnt, post, msg, ldmax);
        else
          Launch<T>::template hl<false>(impl_.rule, lreg, t, llds, s, g, st, ltab, cnt, post, msg, ldmax);
        timed_end(kKernelLayer, s);
      }
      // horizontal_layered.rs:66-78
      pack(post);
      syndrome_of(w.hardbits, w.unsat0);
      latch(w.unsat0, static_cast<int32_t>(it));
      if (checkpoint_due(it)) compact(max_iterations - it, msg, false, hl_rec ? m * 3 : static_cast<uint32_t>(e_));
    }
  }

  emit(zero_fill, 0);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- one group of codewords, 8-bit quantised arithmetics (kernels_i8.hip.h) ------------------

int DeviceDecoder::run_group_i8(Workspace &w, const void *llrs, bool llrs_f64, size_t nb, uint32_t max_iterations,
                                uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior, hipStream_t s,
                                bool may_block) {
  const uint32_t G = static_cast<uint32_t>(w.G);
  const uint32_t W = G / 64, tile = 256;
  const uint32_t n = static_cast<uint32_t>(n_), m = static_cast<uint32_t>(m_);
  int8_t *chan = static_cast<int8_t *>(w.chan), *msg = static_cast<int8_t *>(w.msg);
  int16_t *post = static_cast<int16_t *>(w.post);
  const uint32_t target_waves = opt_waves_ ? opt_waves_ : 128 * 1024;
  dev::Graph g{d_row_ptr_, d_edge_col_, d_col_ptr_, d_col_edge_, m, n, static_cast<uint32_t>(e_),
               nullptr,    nullptr,     nullptr,    0,           nullptr, nullptr};
  dev::State st{w.done, w.iters, w.n_active, w.n_slots, w.slot_cw, nullptr, 0, 0, nullptr, nullptr, 0};
  // progress word: the first check-node launch of iteration `it` runs with ticked(it)
  w.epoch = (w.epoch % 0xFFFFFFu) + 1;
  auto ticked = [&](uint32_t it) {
    dev::State t = st;
    t.publish = opt_poll_ ? w.d_flag : nullptr;
    t.epoch = w.epoch;
    t.tick = it;
    return t;
  };
  const ProgressPoll poll{(opt_poll_ && w.d_flag) ? w.h_flag : nullptr, w.epoch, may_block,
                          t_pace_lead ? t_pace_lead : (impl_.schedule == Schedule::Layered ? 2u : 8u), s};
  const dev::I8Opts o{impl_.rule == Rule::Aminstar, impl_.jones, impl_.hardlimit, impl_.deg1clip};

  dev::init_group_kernel<<<(G + 255) / 256, 256, 0, s>>>(w.done, w.iters, w.unsat0, w.unsat1, w.n_active, w.n_slots,
                                                         w.slot_cw, static_cast<uint32_t>(nb), G);
  {
    dim3 grid((n + 63) / 64, W);
    const uint32_t block_size = pattern_len_ ? n / pattern_len_ : 0;
    if (llrs_f64)
      dev::ingest_i8_kernel<double><<<grid, 256, 0, s>>>(static_cast<const double *>(llrs), input_len_,
                                                        static_cast<uint32_t>(nb), n, G, tile, chan, post, w.rawbits,
                                                        d_src_block_, block_size);
    else
      dev::ingest_i8_kernel<float><<<grid, 256, 0, s>>>(static_cast<const float *>(llrs), input_len_,
                                                       static_cast<uint32_t>(nb), n, G, tile, chan, post, w.rawbits,
                                                       d_src_block_, block_size);
    if (w.after_ingest) {
      HIP_TRY(hipEventRecord(w.after_ingest, s));
      if (w.ingest_seq) w.ingest_seq->fetch_add(1, std::memory_order_release);
    }
  }
  // a wavefront takes 64 packed words of a few checks; enough wavefronts to fill the chip
  const uint32_t synd_chunks = (W + 63) / 64;
  const uint32_t synd_rows =
      std::max<uint32_t>(1, std::min<uint32_t>(64, uint32_t(uint64_t(m) * synd_chunks * 64 / opt_synd_threads_)));
  const uint32_t synd_threads = 64 * synd_chunks * ((m + synd_rows - 1) / synd_rows);
  auto syndrome_of = [&](const uint64_t *hard, uint32_t *unsat) {
    if (m == 0) return;
    dev::syndrome_bits_kernel<<<(synd_threads + 255) / 256, 256, 0, s>>>(d_row_ptr_, d_edge_col_, m, hard, unsat,
                                                                         w.n_active, w.n_slots, W, synd_rows);
  };
  auto latch = [&](uint32_t *unsat, int32_t it) {
    dev::latch_kernel<<<(G + 255) / 256, 256, 0, s>>>(w.done, w.iters, unsat, w.n_active, it, G);
  };
  const Tiling pack_t = make_tiling(G, tile, 128, n, 256, target_waves);
  auto pack = [&]() {
    dev::pack_hard_pair_kernel<int16_t><<<pack_t.blocks, pack_t.threads, 0, s>>>(post, w.hardbits, w.n_active, w.n_slots,
                                                                                n, tile, W, pack_t.sched.waves_per_chunk);
  };
  syndrome_of(w.rawbits, w.unsat0);
  latch(w.unsat0, 0);

  uint32_t threads = 256;
  size_t lds = 0;
  // rows beyond the LDS (more than 320 edges): the columns live in HBM, one region per wavefront of a small launch
  const bool i8_fits = staged_block(2, max_row_weight_, 4, &threads, &lds) && lds + 32 <= 160 * 1024;
  if (!i8_fits) {
    threads = kScratchThreads;
    lds = 0;
    const size_t waves_bound = size_t(kScratchWaves) + size_t(G / 256) * (kScratchThreads / 64);
    if (int rc = ensure_row_scratch(w, waves_bound * 2 * max_row_weight_ * 64 * 4)) return rc;
  }
  uint32_t *const i8_scratch = static_cast<uint32_t *>(w.row_scratch);
  lds += 32;  // the correction lookup table (kernels_i8.hip.h, i8_table_init)
  auto set_lds = [&](const void *k) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  };
  uint32_t *unsat[2] = {w.unsat0, w.unsat1};
  int zero_fill = 0;
  if (impl_.schedule == Schedule::Flooding) {
    const Tiling cn_t = make_tiling(G, tile, 256, m, threads, i8_fits ? target_waves : std::min(target_waves, kScratchWaves));
    const Tiling vn_t = make_tiling(G, tile, 256, n, 256, target_waves);
    if (!i8_fits && scratch_bytes_for(cn_t, max_row_weight_, 4) > w.row_scratch_bytes) {
      fail("internal error: row scratch smaller than the check-node launch");
      return -3;
    }
    set_lds(reinterpret_cast<const void *>(dev::cn_i8_kernel<true>));
    set_lds(reinterpret_cast<const void *>(dev::cn_i8_kernel<false>));
    for (uint32_t it = 1; it <= max_iterations; it++) {
      if (it > 1 && poll.finished(it)) break;
      const bool first = it == 1;
      uint32_t *unsat_out = unsat[it & 1];
      const dev::State stp = ticked(it);
      timed_begin(kKernelCheck, s);
      if (!i8_fits) {
        if (first)
          dev::cn_i8_kernel<true, true><<<cn_t.blocks, cn_t.threads, 0, s>>>(g, cn_t.sched, stp, o, chan, post, msg, unsat_out,
                                                                             max_row_weight_, i8_scratch);
        else
          dev::cn_i8_kernel<false, true><<<cn_t.blocks, cn_t.threads, 0, s>>>(g, cn_t.sched, stp, o, chan, post, msg, unsat_out,
                                                                              max_row_weight_, i8_scratch);
      } else if (first)
        dev::cn_i8_kernel<true><<<cn_t.blocks, cn_t.threads, lds, s>>>(g, cn_t.sched, stp, o, chan, post, msg, unsat_out,
                                                                       max_row_weight_);
      else
        dev::cn_i8_kernel<false><<<cn_t.blocks, cn_t.threads, lds, s>>>(g, cn_t.sched, stp, o, chan, post, msg,
                                                                        unsat_out, max_row_weight_);
      timed_end(kKernelCheck, s);
      timed_begin(kKernelVar, s);
      dev::vn_i8_kernel<<<vn_t.blocks, vn_t.threads, 0, s>>>(g, vn_t.sched, st, o, chan, msg, post,
                                                             first ? nullptr : unsat_out, unsat[(it + 1) & 1],
                                                             static_cast<int32_t>(it) - 1);
      timed_end(kKernelVar, s);
    }
    if (max_iterations > 0) {
      pack();
      uint32_t *u = unsat[(max_iterations + 1) & 1];
      syndrome_of(w.hardbits, u);
      latch(u, static_cast<int32_t>(max_iterations));
    } else {
      zero_fill = 1;
    }
  } else {
    const uint32_t n_levels = level_ptr_.empty() ? 0 : static_cast<uint32_t>(level_ptr_.size() - 1);
    const dev::State st0 = st;
    set_lds(reinterpret_cast<const void *>(dev::hl_i8_kernel<true>));
    set_lds(reinterpret_cast<const void *>(dev::hl_i8_kernel<false>));
    const bool serial = n_levels > opt_serial_levels_;
    const uint32_t n_launch = serial ? std::min<uint32_t>(n_levels, 1) : n_levels;
    for (uint32_t it = 1; it <= max_iterations; it++) {
      if (it > 1 && poll.finished(it)) break;
      const dev::State stp = ticked(it);
      for (uint32_t l = 0; l < n_launch; l++) {
        const dev::State &st = l == 0 ? stp : st0;
        const uint32_t r0 = serial ? 0 : level_ptr_[l], cnt = serial ? m : level_ptr_[l + 1] - level_ptr_[l];
        // per level: LDS columns as tall as this level's longest row; register-resident rows when short
        const uint32_t ldmax = std::max<uint32_t>(serial ? max_row_weight_ : level_maxdeg_[l], 1);
        uint32_t lthreads = threads;
        size_t llds = 0;
        bool lfits = staged_block(2, ldmax, 4, &lthreads, &llds);
        if (serial) {
          lthreads = 64;          // row-serial mode (see run_group): one wave per 256-codeword slice
          llds = size_t(2) * ldmax * 64 * 4;
        }
        llds += 32;
        lfits = lfits && llds <= 160 * 1024;
        if (!lfits) {
          lthreads = serial ? 64 : kScratchThreads;
          llds = 0;
        }
        const uint32_t lreg = !opt_hl_reg_ ? 0 : (ldmax <= 12 ? 12 : (ldmax <= 24 ? 24 : 0));
        const Tiling t = make_tiling(G, tile, 256, serial ? 1 : cnt, lthreads, lfits ? target_waves : std::min(target_waves, kScratchWaves));
        if (!lfits && scratch_bytes_for(t, ldmax, 4) > w.row_scratch_bytes) {
          fail("internal error: row scratch smaller than a level's launch");
          return -3;
        }
        auto launch = [&](auto k) {
          if (llds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(llds));
          k<<<t.blocks, t.threads, llds, s>>>(g, t.sched, st, o, d_level_rows_ + r0, cnt, post, msg, ldmax);
        };
        auto launch_staged = [&](auto k, uint32_t *scratch) {
          if (llds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(llds));
          k<<<t.blocks, t.threads, llds, s>>>(g, t.sched, st, o, d_level_rows_ + r0, cnt, post, msg, ldmax, scratch);
        };
        timed_begin(kKernelLayer, s);
        if (it == 1) {
          if (lreg == 12)
            launch(dev::hl_i8_reg_kernel<12, true>);
          else if (lreg == 24)
            launch(dev::hl_i8_reg_kernel<24, true>);
          else if (!lfits)
            launch_staged(dev::hl_i8_kernel<true, true>, i8_scratch);
          else
            launch_staged(dev::hl_i8_kernel<true>, nullptr);
        } else {
          if (lreg == 12)
            launch(dev::hl_i8_reg_kernel<12, false>);
          else if (lreg == 24)
            launch(dev::hl_i8_reg_kernel<24, false>);
          else if (!lfits)
            launch_staged(dev::hl_i8_kernel<false, true>, i8_scratch);
          else
            launch_staged(dev::hl_i8_kernel<false>, nullptr);
        }
        timed_end(kKernelLayer, s);
      }
      pack();
      syndrome_of(w.hardbits, w.unsat0);
      latch(w.unsat0, static_cast<int32_t>(it));
    }
  }
  {
    dim3 grid(std::min<uint32_t>((n + 63) / 64, 4096), W);
    if (llrs_f64)
      dev::emit_kernel<int16_t, double><<<grid, 256, 0, s>>>(post, w.rawbits, st, nullptr, n, G, tile,
                                                            static_cast<uint32_t>(out_len), bits, iterations,
                                                            static_cast<double *>(posterior), zero_fill, 0);
    else
      dev::emit_kernel<int16_t, float><<<grid, 256, 0, s>>>(post, w.rawbits, st, nullptr, n, G, tile,
                                                           static_cast<uint32_t>(out_len), bits, iterations,
                                                           static_cast<float *>(posterior), zero_fill, 0);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int DeviceDecoder::run_any(Workspace &w, const void *llrs, bool llrs_f64, size_t nb, uint32_t max_iterations,
                           uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior, hipStream_t s,
                           bool may_block, bool own_thread) {
  // The host may only wait on the device's progress where the call is synchronous anyway, and it
  // pays only for small groups of the layered schedule (dozens of short launches per iteration);
  // with flooding's two launches per iteration waiting costs more than the empty launches it saves,
  // and large groups keep the host free to fill both lanes -- unless every lane has its own enqueuing thread.
  const bool may_wait = may_block;
  may_block = may_block && opt_poll_ && impl_.schedule == Schedule::Layered && (own_thread || nb * n_ <= size_t(8) * 1000 * 1000);
  // A lane's own enqueuing thread always paces itself, one iteration ahead: the call cannot return before its threads
  // have enqueued everything anyway, and the command queue lets a thread run about four iterations ahead -- with early
  // termination that is four iterations of launches that return at once (35 each on 5G NR BG1, 5 us apiece) behind the
  // last real one.  Config 3 at +2 dB: 346 k -> 351 k codewords/s; fixed work unchanged (round 4).
  t_pace_lead = opt_lead_;
  // (Only where the call may wait at all -- the library's own stream, host buffers, or option "throttle": a call that
  // merely enqueues on the CALLER's stream returns as soon as everything is enqueued, as include/ldpc_toolbox.h says.)
  if (own_thread && may_wait && opt_poll_ && impl_.schedule == Schedule::Layered && opt_lane_pace_) {
    may_block = true;
    if (!opt_lead_) t_pace_lead = 1;
  }
  // Flooding where the call may wait -- the library's own stream (a NULL stream: synchronous anyway) or option "throttle"
  // (round 5; the simulation driver sets it): the host follows the group two iterations
  // ahead -- every flooding check-node kernel publishes the progress word -- which costs the device nothing (two iterations
  // are several milliseconds of queued work) and lets the host SEE convergence begin: from then on it asks for a re-packing
  // checkpoint after every iteration instead of every second one (run_group: +1.1 % at config 2's +2 dB, and not one extra
  // launch in a call where nothing converges), and it stops enqueuing with the group.
  // (One-lane calls of the device-resident entry, or a lane with an enqueuing thread of its own: one thread enqueuing both
  // lanes' groups in turn must not wait on the first; the host-buffer entry's calling thread stages the next group's copy
  // between its enqueues.)
  if (impl_.schedule == Schedule::Flooding && !impl_.i8 && opt_poll_ && may_wait && !profiling_ && (t_flood_pace || own_thread)) {
    may_block = true;
    if (!opt_lead_) t_pace_lead = 2;
  }
  return impl_.i8 ? run_group_i8(w, llrs, llrs_f64, nb, max_iterations, bits, out_len, iterations, posterior, s, may_block)
         : impl_.f64
             ? run_group<double>(w, llrs, llrs_f64, nb, max_iterations, bits, out_len, iterations, posterior, s, may_block)
             : run_group<float>(w, llrs, llrs_f64, nb, max_iterations, bits, out_len, iterations, posterior, s, may_block);
}

// The handle's streams are non-blocking, so the legacy default stream (handle 0: what a caller that
// never made a stream works on -- torch's default stream is that one) does not order them.  A call
// that lets the library pick the stream (hip_stream == NULL) is therefore ordered explicitly after
// whatever the default stream holds at this moment: the caller's buffers may still be being written
// there.
int DeviceDecoder::order_after_default_stream(hipStream_t s) {
  HIP_TRY(hipEventRecord(ev_default_, nullptr));
  HIP_TRY(hipStreamWaitEvent(s, ev_default_, 0));
  return 0;
}

int DeviceDecoder::decode_device(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations,
                                 uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior,
                                 hipStream_t stream) {
  if (batch == 0) return 0;
  if (out_len > n_) {
    fail("output_len larger than the codeword length");
    return -1;
  }
  HIP_TRY(hipSetDevice(device_));
  const bool own_stream = stream == nullptr;
  hipStream_t s = own_stream ? stream_ : stream;
  if (own_stream)
    if (int rc = order_after_default_stream(s)) return rc;
  // the single-launch path needs every one of its workgroups resident at once: only calls that return
  // synchronised take it (one at a time per process, see decode_latency); a call that merely enqueues on
  // the caller's stream keeps the batched kernels
  if (lat_ && batch <= opt_latency_ && own_stream) {
    const int rc = decode_latency(llrs, llrs_f64, false, batch, max_iterations, bits, out_len, iterations, posterior, s);
    if (rc != kLatencyRetry) return rc;
  }
  if (lat_edge_ && batch <= edge_latency_limit() && own_stream) {
    const int rc = decode_latency_edge(llrs, llrs_f64, false, batch, max_iterations, bits, out_len, iterations, posterior, s);
    if (rc != kLatencyRetry) return rc;
  }
  size_t G = pick_group(batch);
  uint32_t lanes = lane_count();
  // a batch that fits one group is split in two halves when each half's launches still fill the
  // chip (small codes lose more from the thinner launches than the overlap returns)
  if (lanes == 2 && batch <= G && split_pays(batch)) G = round_up((batch + 1) / 2, 256);
  if (batch <= G) lanes = 1;
  last_lanes_ = lanes;
  last_group_ = G;
  if (int rc = ensure_lanes(lanes, G)) return rc;
  if (lanes == 2) {
    HIP_TRY(hipEventRecord(ev_fork_, s));
    HIP_TRY(hipStreamWaitEvent(stream2_, ev_fork_, 0));
  }
  const size_t in_elem = llrs_f64 ? 8 : 4;
  // The layered schedule enqueues dozens of launches per iteration and polls the group's progress word between
  // iterations: with one host thread the second lane's launches were only enqueued once the first lane's whole
  // iteration sequence had been (the lanes then overlap only when every group runs all its iterations; with early
  // termination they ran one after the other, followed by the launches enqueued past convergence).  Each lane gets
  // its own enqueuing thread, and may then wait on its own progress (run_any).
  // (Flooding with two lanes under option "throttle", round 5: a thread per lane too, so that each lane's host side can follow its
  // own group -- the paced host of run_any with its tail checkpoints -- instead of one thread enqueuing both groups blind.)
  const bool threaded = lanes == 2 && opt_lane_threads_ && !profiling_ && max_iterations > 0 &&
                        (impl_.schedule == Schedule::Layered ||
                         (impl_.schedule == Schedule::Flooding && !impl_.i8 && (opt_throttle_ || own_stream) && opt_poll_));
  const bool may_block = own_stream || opt_throttle_;
  t_flood_pace = lanes == 1;
  auto run_groups = [&](uint32_t only_lane) -> int {  // only_lane: 0 / 1 = that lane's groups, 2 = all of them in turn
    uint32_t gi = 0;
    for (size_t b0 = 0; b0 < batch; b0 += G, gi++) {
      const uint32_t lane = lanes == 2 ? (gi & 1u) : 0u;
      if (only_lane != 2 && lane != only_lane) continue;
      const size_t nb = std::min(G, batch - b0);
      const char *src = static_cast<const char *>(llrs) + b0 * input_len_ * in_elem;
      uint8_t *dst_bits = bits + b0 * out_len;
      int32_t *dst_it = iterations ? iterations + b0 : nullptr;
      void *dst_post = posterior ? static_cast<char *>(posterior) + b0 * n_ * in_elem : nullptr;
      const bool skew = !threaded && lanes == 2 && opt_lane_skew_ && impl_.schedule == Schedule::Flooding && max_iterations > 0;
      if (skew && gi == 0) skew_record_ = ev_skew_;
      if (skew && gi == 1) HIP_TRY(hipStreamWaitEvent(stream2_, ev_skew_, 0));
      if (int rc = run_any(*ws_[lane], src, llrs_f64, nb, max_iterations, dst_bits, out_len, dst_it, dst_post,
                           lane ? stream2_ : s, may_block, threaded))
        return rc;
      if (skew) skew_record_ = nullptr;  // (never with lane threads: the field belongs to the calling thread)
    }
    return 0;
  };
  if (threaded) {
    int rc1 = 0;
    std::thread second([&] {
      rc1 = hipSetDevice(device_) == hipSuccess ? run_groups(1) : -2;
    });
    const int rc0 = run_groups(0);
    second.join();
    if (rc0 || rc1) return rc0 ? rc0 : rc1;
  } else if (int rc = run_groups(2)) {
    return rc;
  }
  if (lanes == 2) {
    HIP_TRY(hipEventRecord(ev_join_, stream2_));
    HIP_TRY(hipStreamWaitEvent(s, ev_join_, 0));
  }
  if (own_stream) HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int DeviceDecoder::ensure_row_scratch(Workspace &w, size_t bytes) {
  if (w.row_scratch_bytes >= bytes) return 0;
  if (w.row_scratch) (void)hipFree(w.row_scratch);   // (waits for the work that may still use it)
  w.row_scratch = nullptr;
  w.row_scratch_bytes = 0;
  HIP_TRY(hipMalloc(&w.row_scratch, bytes));
  w.row_scratch_bytes = bytes;
  return 0;
}

int DeviceDecoder::ensure_host_staging(Workspace &w, size_t G, size_t in_elem) {
  const size_t in_bytes = G * input_len_ * in_elem;
  if (w.in_bytes < in_bytes) {
    if (w.in) (void)hipFree(w.in);
    w.in = nullptr;
    w.in_bytes = 0;
    HIP_TRY(hipMalloc(&w.in, in_bytes));
    w.in_bytes = in_bytes;
  }
  return 0;
}

// ---- host-pointer entry --------------------------------------------------------------------
// What a C caller of the batch entries hands over is pageable memory.  A hipMemcpy from pageable
// memory stages through the runtime's own bounce buffers and blocks the calling thread, so the copy of
// group g+1 could not be queued while the launches of group g were being enqueued.  The library
// stages itself instead:
//   * two rings of pinned chunks (hipHostMalloc), one per direction; the calling thread copies the
//     caller's rows into a chunk with a few threads, then a DMA on a copy stream of its own moves the
//     chunk to the lane's device input buffer -- the DMA of chunk c overlaps the memcpy of chunk c+1
//     and the decode of the previous groups;
//   * a lane's input buffer is free again as soon as its group has been INGESTED (an event recorded
//     right after the ingest kernel), not when it has been decoded, so staging runs a full group ahead;
//   * the results of the whole batch stay in device buffers and are drained at the end, group by group
//     as each completes (the last group's drain is all that is exposed);
//   * the first two groups are a quarter and three quarters of a group, so that decoding starts after
//     a quarter of a group has crossed the bus.
namespace {

// memcpy with a few threads: one core moves about 10 GB/s, the bus several times that
void par_memcpy(char *dst, const char *src, size_t bytes, unsigned threads) {
  if (threads <= 1 || bytes < (size_t(4) << 20)) {
    std::memcpy(dst, src, bytes);
    return;
  }
  const size_t per = round_up((bytes + threads - 1) / threads, 4096);
  std::vector<std::thread> th;
  for (size_t off = per; off < bytes; off += per)
    th.emplace_back([=] { std::memcpy(dst + off, src + off, std::min(per, bytes - off)); });
  std::memcpy(dst, src, std::min(per, bytes));
  for (auto &t : th) t.join();
}

}  // namespace

int DeviceDecoder::ensure_pipe(size_t group, size_t out_len, size_t in_elem, bool posterior) {
  if (!pipe_) {
    // built aside and published only when complete: a half-built pipe must never be seen by a later call
    HostPipe *p = new HostPipe();
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    p->copy_threads = std::min(8u, std::max(1u, hw / 2));
    bool ok = hipStreamCreateWithFlags(&p->h2d, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&p->d2h, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < HostPipe::kSlots; i++)
      ok = hipEventCreateWithFlags(&p->in_done[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&p->out_done[i], hipEventDisableTiming) == hipSuccess;
    for (int l = 0; ok && l < 2; l++)
      ok = hipEventCreateWithFlags(&p->in_ready[l], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&p->ingested[l], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      p->release();
      delete p;
      fail("host staging: stream / event creation failed");
      return -2;
    }
    pipe_ = p;
  }
  HostPipe &p = *pipe_;
  auto grow = [&](void **ptr, size_t *have, size_t want) -> int {
    if (*have >= want) return 0;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(ptr, want));
    *have = want;
    return 0;
  };
  // results stay on the device for one group at a time per ring entry (not for the whole batch: a long call
  // with posteriors would not fit), drained while later groups decode
  for (int r = 0; r < HostPipe::kOutRing; r++) {
    if (int rc = grow(reinterpret_cast<void **>(&p.d_bits[r]), &p.bits_cap[r], std::max<size_t>(group * out_len, 1))) return rc;
    if (int rc = grow(reinterpret_cast<void **>(&p.d_iters[r]), &p.iters_cap[r], group * sizeof(int32_t))) return rc;
    if (posterior)
      if (int rc = grow(&p.d_post[r], &p.post_cap[r], group * n_ * in_elem)) return rc;
  }
  return 0;
}

// caller's (pageable) memory -> device, through the pinned ring, on the h2d stream
int DeviceDecoder::stage_in(const char *src, char *dst, size_t bytes) {
  HostPipe &p = *pipe_;
  for (size_t off = 0; off < bytes; off += HostPipe::kChunk) {
    const size_t len = std::min(HostPipe::kChunk, bytes - off);
    const int slot = p.next_in;
    p.next_in = (p.next_in + 1) % HostPipe::kSlots;
    HIP_TRY(hipEventSynchronize(p.in_done[slot]));  // the DMA that last used this chunk has finished
    if (HostPipe::pinned(&p.in_slot[slot], &p.in_cap[slot], len)) {
      fail("host staging: pinned input chunk");
      return -2;
    }
    par_memcpy(p.in_slot[slot], src + off, len, p.copy_threads);
    HIP_TRY(hipMemcpyAsync(dst + off, p.in_slot[slot], len, hipMemcpyHostToDevice, p.h2d));
    HIP_TRY(hipEventRecord(p.in_done[slot], p.h2d));
  }
  return 0;
}

// device -> caller's memory, through the pinned ring, on the d2h stream (which the caller has already
// ordered after the producer); up to kSlots DMAs in flight ahead of the host-side copies
int DeviceDecoder::drain_out(char *dst, const char *src, size_t bytes) {
  HostPipe &p = *pipe_;
  const size_t chunks = (bytes + HostPipe::kChunk - 1) / HostPipe::kChunk;
  auto issue = [&](size_t c) -> int {
    const size_t off = c * HostPipe::kChunk, len = std::min(HostPipe::kChunk, bytes - off);
    const int slot = static_cast<int>(c % HostPipe::kSlots);
    if (HostPipe::pinned(&p.out_slot[slot], &p.out_cap[slot], len)) {
      fail("host staging: pinned output chunk");
      return -2;
    }
    HIP_TRY(hipMemcpyAsync(p.out_slot[slot], src + off, len, hipMemcpyDeviceToHost, p.d2h));
    HIP_TRY(hipEventRecord(p.out_done[slot], p.d2h));
    return 0;
  };
  for (size_t c = 0; c < std::min<size_t>(chunks, HostPipe::kSlots); c++)
    if (int rc = issue(c)) return rc;
  for (size_t c = 0; c < chunks; c++) {
    const size_t off = c * HostPipe::kChunk, len = std::min(HostPipe::kChunk, bytes - off);
    const int slot = static_cast<int>(c % HostPipe::kSlots);
    HIP_TRY(hipEventSynchronize(p.out_done[slot]));
    par_memcpy(dst + off, p.out_slot[slot], len, p.copy_threads);
    if (c + HostPipe::kSlots < chunks)
      if (int rc = issue(c + HostPipe::kSlots)) return rc;
  }
  return 0;
}

int DeviceDecoder::decode_host(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations,
                               uint8_t *bits, size_t out_len, int32_t *iterations, void *posterior) {
  if (batch == 0) return 0;
  t_flood_pace = false;  // (this thread stages copies between its enqueues: it does not wait on a flooding group)
  if (out_len > n_) {
    fail("output_len larger than the codeword length");
    return -1;
  }
  HIP_TRY(hipSetDevice(device_));
  if (lat_ && batch <= opt_latency_) {
    const int rc = decode_latency(llrs, llrs_f64, true, batch, max_iterations, bits, out_len, iterations, posterior, stream_);
    if (rc != kLatencyRetry) return rc;  // else: its workgroups could not all become resident -> batched kernels
  }
  if (lat_edge_ && batch <= edge_latency_limit()) {
    const int rc = decode_latency_edge(llrs, llrs_f64, true, batch, max_iterations, bits, out_len, iterations, posterior, stream_);
    if (rc != kLatencyRetry) return rc;
  }
  size_t G = pick_group(batch);
  // (host buffers go through pinned staging rings sized by the group: keep a ring slot within 256 MB)
  while (!group_pref_ && G > 4096 && G * input_len_ * (llrs_f64 ? 8 : 4) > (size_t(256) << 20)) G /= 2;
  if (lane_count() == 2 && batch <= G && split_pays(batch)) G = round_up((batch + 1) / 2, 256);
  const uint32_t lanes = (batch > G && opt_lanes_ != 1) ? 2u : 1u;
  last_lanes_ = lanes;
  last_group_ = G;
  const size_t in_elem = llrs_f64 ? 8 : 4;
  if (int rc = ensure_lanes(lanes, G)) return rc;
  for (uint32_t l = 0; l < lanes; l++)
    if (int rc = ensure_host_staging(*ws_[l], G, in_elem)) return rc;
  if (int rc = ensure_pipe(std::min(G, batch), out_len, in_elem, posterior != nullptr)) return rc;
  HostPipe &p = *pipe_;
  // group boundaries: a long batch opens with G/4 and 3G/4 (decoding starts after a quarter group's copy)
  std::vector<size_t> starts;
  {
    // (one execution lane too: its groups run one after the other, but the next group's copy overlaps the current group's
    // decode all the same, and the FIRST copy overlaps nothing -- a quarter group's copy is a quarter of that exposure;
    // "host_split" = 0 keeps whole groups)
    size_t b0 = 0;
    if ((lanes == 2 || opt_host_split_) && batch >= 2 * G && G >= 1024 && (G / 4) % 256 == 0) {
      starts.push_back(0);
      starts.push_back(G / 4);
      b0 = G;
    }
    for (; b0 < batch; b0 += G) starts.push_back(b0);
    // ... and closes with a short group: the last group's results are the only ones whose way back is exposed
    const size_t last0 = starts.back(), last_n = batch - last0;
    if ((lanes == 2 || opt_host_split_) && starts.size() >= 3 && last_n >= 1024) {
      const size_t tail = std::max<size_t>(256, last_n / 4 / 256 * 256);
      starts.push_back(batch - tail);
    }
    starts.push_back(batch);
  }
  const size_t n_groups = starts.size() - 1;
  while (p.group_done.size() < n_groups) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    p.group_done.push_back(e);
  }
  hipStream_t streams[2] = {stream_, stream2_};
  const size_t row_in = input_len_ * in_elem;
  constexpr size_t R = HostPipe::kOutRing;
  int rc = 0;
  // results of group gi: device ring entry gi % R -> the caller's rows (blocks until the group has finished;
  // the groups queued behind it keep the device busy meanwhile)
  auto drain_group = [&](size_t gi) -> int {
    const size_t b0 = starts[gi], nb = starts[gi + 1] - b0, r = gi % R;
    HIP_TRY(hipStreamWaitEvent(p.d2h, p.group_done[gi], 0));
    int drc = 0;
    if (out_len) drc = drain_out(reinterpret_cast<char *>(bits + b0 * out_len), reinterpret_cast<const char *>(p.d_bits[r]), nb * out_len);
    if (drc == 0 && iterations)
      drc = drain_out(reinterpret_cast<char *>(iterations + b0), reinterpret_cast<const char *>(p.d_iters[r]), nb * sizeof(int32_t));
    if (drc == 0 && posterior)
      drc = drain_out(static_cast<char *>(posterior) + b0 * n_ * in_elem, static_cast<const char *>(p.d_post[r]), nb * n_ * in_elem);
    return drc;
  };
  // The layered schedule enqueues dozens of launches per iteration: each lane's launches are enqueued by a thread of its
  // own (decode_device: same reason), which may then pace itself on the group's progress word, while this thread goes
  // on staging the next group.  Host-side order between the threads where an event is recorded by one and waited on
  // by the other: the counters below (a wait enqueued before the record would not wait at all).
  const bool threaded = lanes == 2 && n_groups >= 2 && impl_.schedule == Schedule::Layered && opt_lane_threads_ && !profiling_ &&
                        max_iterations > 0;
  struct LaneQueue {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<int()>> tasks;
    bool closing = false;
    std::atomic<uint32_t> ingest_recorded{0}, enqueued{0};
    std::atomic<int> rc{0};
  };
  LaneQueue lq[2];
  auto wait_for = [&](std::atomic<uint32_t> &counter, uint32_t at_least, LaneQueue &q) {
    while (counter.load(std::memory_order_acquire) < at_least && q.rc.load() == 0) std::this_thread::yield();
  };
  if (threaded) {
    for (uint32_t l = 0; l < 2; l++) {
      lq[l].th = std::thread([this, l, &lq] {
        LaneQueue &q = lq[l];
        if (hipSetDevice(device_) != hipSuccess) q.rc = -2;
        for (;;) {
          std::function<int()> task;
          {
            std::unique_lock<std::mutex> lock(q.m);
            q.cv.wait(lock, [&] { return q.closing || !q.tasks.empty(); });
            if (q.tasks.empty()) return;
            task = std::move(q.tasks.front());
            q.tasks.pop_front();
          }
          if (q.rc.load() == 0) {
            const int trc = task();
            if (trc) q.rc = trc;
          }
          q.enqueued.fetch_add(1, std::memory_order_release);
        }
      });
    }
  }
  auto close_lanes = [&]() {
    if (!threaded) return;
    for (auto &q : lq) {
      {
        std::lock_guard<std::mutex> lock(q.m);
        q.closing = true;
      }
      q.cv.notify_all();
      if (q.th.joinable()) q.th.join();
      if (q.rc.load() && rc == 0) rc = q.rc.load();
    }
  };
  // Every way out of this function -- the error returns included -- joins the lane threads, takes the workspaces'
  // pointers to this call's stack objects back, and leaves every stream of the call idle (nothing may still read the
  // caller's rows or write into a counter that no longer exists).
  bool finished = false;
  auto finish = [&]() {
    if (finished) return;
    finished = true;
    close_lanes();
    for (uint32_t l = 0; l < lanes; l++) {
      ws_[l]->after_ingest = nullptr;
      ws_[l]->ingest_seq = nullptr;
    }
    for (hipStream_t st : {p.h2d, streams[0], streams[1], p.d2h}) {
      const hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess && rc == 0) {
        fail("hipStreamSynchronize", e);
        rc = -2;
      }
    }
  };
  struct AtExit {
    std::function<void()> f;
    ~AtExit() { f(); }
  } at_exit{finish};
  // (threaded: group gi's launches and its group_done record have been made)
  auto enqueued = [&](size_t gi) {
    if (threaded) wait_for(lq[gi % lanes].enqueued, static_cast<uint32_t>(gi / lanes + 1), lq[gi % lanes]);
  };
  auto hip_ok = [&](hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fail(what, e);
    rc = -2;
    return false;
  };
  size_t drained = 0;
  for (size_t gi = 0; gi < n_groups && rc == 0; gi++) {
    const size_t b0 = starts[gi], nb = starts[gi + 1] - b0;
    const uint32_t lane = static_cast<uint32_t>(gi % lanes);
    Workspace &w = *ws_[lane];
    hipStream_t s = streams[lane];
    // this group's ring entry must have been drained
    if (gi >= R) {
      enqueued(gi - R);
      rc = drain_group(gi - R);
      drained = gi - R + 1;
      if (rc) break;
    }
    // the lane's input buffer: free once the lane's previous group has been ingested
    if (gi >= lanes) {
      if (threaded) wait_for(lq[lane].ingest_recorded, static_cast<uint32_t>(gi / lanes), lq[lane]);
      if (!hip_ok(hipStreamWaitEvent(p.h2d, p.ingested[lane], 0), "hipStreamWaitEvent")) break;
    }
    rc = stage_in(static_cast<const char *>(llrs) + b0 * row_in, static_cast<char *>(w.in), nb * row_in);
    if (rc) break;
    if (!hip_ok(hipEventRecord(p.in_ready[lane], p.h2d), "hipEventRecord")) break;
    const size_t r = gi % R;
    hipEvent_t ingested = p.ingested[lane];
    std::atomic<uint32_t> *ingest_seq = threaded ? &lq[lane].ingest_recorded : nullptr;
    // (the workspace's "record this after the ingest" fields are written by whoever enqueues the lane's launches --
    // the lane's own thread, or this one when there are none -- never by one thread while another reads them)
    auto enqueue = [=, &w, &p]() -> int {
      w.after_ingest = ingested;
      w.ingest_seq = ingest_seq;
      HIP_TRY(hipStreamWaitEvent(s, p.in_ready[lane], 0));
      // a single small group (the reference-style scalar call) may let the host follow the device's progress; so may
      // a lane with an enqueuing thread of its own
      const int erc = run_any(w, w.in, llrs_f64, nb, max_iterations, p.d_bits[r], out_len, p.d_iters[r],
                              posterior ? p.d_post[r] : nullptr, s, n_groups == 1 || threaded, threaded);
      if (erc) return erc;
      HIP_TRY(hipEventRecord(p.group_done[gi], s));
      return 0;
    };
    if (threaded) {
      {
        std::lock_guard<std::mutex> lock(lq[lane].m);
        lq[lane].tasks.push_back(enqueue);
      }
      lq[lane].cv.notify_one();
      if (lq[lane].rc.load()) rc = lq[lane].rc.load();
    } else {
      rc = enqueue();
    }
    if (rc) break;
  }
  // the remaining results, group by group as each completes
  for (size_t gi = drained; gi < n_groups && rc == 0; gi++) {
    enqueued(gi);
    if (threaded && lq[gi % lanes].rc.load()) break;
    rc = drain_group(gi);
  }
  finish();
  return rc;
}

// ---- small-batch path ------------------------------------------------------------------------
// One persistent launch decodes the whole (small) batch: latency.hip.h.  host_pointers: the caller's buffers
// are staged through one pinned chunk each way on `s` and the call returns synchronised; else everything is
// device memory and the call only enqueues on `s`.
// The kernel's workgroups synchronise with each other, so all of them must be resident together -- one such
// kernel fills the chip's register files.  Two of them at once (two handles driven by two threads, as the
// reference's BER driver drives its worker threads) would each hold part of the chip and wait for the rest:
// the calls are therefore serialised per process, and always return synchronised.  Should the workgroups
// still not come together (another process's kernels hold CUs for longer than the bounded spins allow),
// the kernel gives up with its error word set and the call is redone by the batched kernels (kLatencyRetry).
static std::mutex g_latency_mutex;

int DeviceDecoder::decode_latency(const void *llrs, bool llrs_f64, bool host_pointers, size_t batch,
                                  uint32_t max_iterations, uint8_t *bits, size_t out_len, int32_t *iterations,
                                  void *posterior, hipStream_t s) {
  std::lock_guard<std::mutex> one_at_a_time(g_latency_mutex);
  LatencyPath &lp = *lat_;
  const size_t in_elem = llrs_f64 ? 8 : 4;
  const uint32_t n = static_cast<uint32_t>(n_), m = static_cast<uint32_t>(m_);
  last_lanes_ = 1;
  last_group_ = batch;
  if (!lp.uploaded) {
    auto up = [&](const std::vector<uint32_t> &v, uint32_t **dst) -> int {
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(v.size(), 1) * sizeof(uint32_t)));
      if (!v.empty()) HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      return 0;
    };
    if (int rc = up(lp.h_rslice_ptr, &lp.d_rslice_ptr)) return rc;
    if (int rc = up(lp.h_rdeg, &lp.d_rdeg)) return rc;
    if (int rc = up(lp.h_col, &lp.d_col)) return rc;
    if (int rc = up(lp.h_vslice_ptr, &lp.d_vslice_ptr)) return rc;
    if (int rc = up(lp.h_vdeg, &lp.d_vdeg)) return rc;
    if (int rc = up(lp.h_vedge, &lp.d_vedge)) return rc;
    if (int rc = up(lp.h_perm, &lp.d_perm)) return rc;
    if (int rc = up(lp.h_inv, &lp.d_inv)) return rc;
    // per-XCD codeword state, each array on a 256-byte boundary (msg: one word per edge id)
    const size_t a_n = round_up((size_t(n) * 2 + 64) * 4, 256), a_m = round_up((size_t(lp.h_rslice_ptr.back()) + 8 * 64) * 4, 256),
                 a_h = round_up(n, 256), slot = 2 * a_n + a_m + a_h;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lp.slots.base), 8 * slot));
    lp.slots.slot_bytes = slot;
    lp.slots.off_post = a_n;
    lp.slots.off_msg = 2 * a_n;
    lp.slots.off_rawhard = 2 * a_n + a_m;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lp.d_sync), sizeof(dev::LatencySync)));
    lp.uploaded = true;
  }
  const void *d_llrs = llrs;
  uint8_t *d_bits = bits;
  int32_t *d_iters = iterations;
  void *d_post = posterior;
  const size_t in_bytes = batch * input_len_ * in_elem, bits_bytes = batch * out_len, post_bytes = batch * n_ * in_elem;
  // The kernel writes the error word into pinned host memory (system scope), and for host-pointer calls it also
  // reads the input there (its ingest: coalesced, in source order, over the bus) and writes the outputs there:
  // a call is memcpy -> one launch -> memcpy with no copy commands (each costs ~10 us of command latency, as
  // much as ten iterations of the decoder; measured -15..25 us per call, profiles/r02_latency.txt).
  const size_t iters_at = round_up(256 + bits_bytes, 256), post_at = round_up(iters_at + batch * sizeof(int32_t), 256);
  const size_t out_need = host_pointers ? post_at + (posterior ? post_bytes : 0) : 256;
  if (lp.pinned(&lp.h_out, &lp.h_out_bytes, out_need) || (host_pointers && lp.pinned(&lp.h_in, &lp.h_in_bytes, in_bytes))) {
    fail("pinned host memory for the small-batch path");
    return -1;
  }
  uint32_t *const o_err = reinterpret_cast<uint32_t *>(lp.h_out);
  *o_err = 0;
  if (host_pointers) {
    std::memcpy(lp.h_in, llrs, in_bytes);
    d_llrs = lp.h_in;
    d_bits = reinterpret_cast<uint8_t *>(lp.h_out + 256);
    d_iters = reinterpret_cast<int32_t *>(lp.h_out + iters_at);
    d_post = posterior ? static_cast<void *>(lp.h_out + post_at) : nullptr;
  }
  HIP_TRY(hipMemsetAsync(lp.d_sync, 0, sizeof(dev::LatencySync), s));
  dev::LatencyTables t{n, m, (m + 63) / 64, (n + 63) / 64, lp.d_rslice_ptr, lp.d_rdeg, lp.d_col, lp.d_vslice_ptr, lp.d_vdeg,
                       lp.d_vedge, lp.d_perm, lp.d_inv, d_src_block_, pattern_len_ ? n / pattern_len_ : 0};
  // one workgroup of 1024 threads per CU, all of them resident together (the kernel's census waits for all of them,
  // and derives how many share an XCD at run time): the grid is what the device can hold at once -- 256 on an
  // MI355X in SPX mode, fewer on a partitioned or smaller device -- and never more than 256
  if (lp.grid == 0) {
    int cus = 0, per_cu_f = 0, per_cu_d = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_f, dev::latency_minsum_kernel<float>, 1024, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_d, dev::latency_minsum_kernel<double>, 1024, 0);
    const int resident = e == hipSuccess ? cus * std::min(per_cu_f, per_cu_d) : 0;
    if (resident < 8) {  // cannot be co-resident in any useful number: this handle keeps the batched kernels
      opt_latency_ = 0;
      opt_latency_edge_ = 0;
      return kLatencyRetry;
    }
    lp.grid = static_cast<uint32_t>(std::min(resident, 256));
  }
  const uint32_t grid = lp.grid;
  if (llrs_f64)
    dev::latency_minsum_kernel<double><<<grid, 1024, 0, s>>>(t, lp.slots, lp.d_sync, static_cast<const double *>(d_llrs),
                                                            static_cast<uint32_t>(input_len_), static_cast<uint32_t>(batch),
                                                            max_iterations, d_bits, static_cast<uint32_t>(out_len), d_iters,
                                                            static_cast<double *>(d_post), o_err LDPC_DBG_ARG(opt_lat_debug_));
  else
    dev::latency_minsum_kernel<float><<<grid, 1024, 0, s>>>(t, lp.slots, lp.d_sync, static_cast<const float *>(d_llrs),
                                                           static_cast<uint32_t>(input_len_), static_cast<uint32_t>(batch),
                                                           max_iterations, d_bits, static_cast<uint32_t>(out_len), d_iters,
                                                           static_cast<float *>(d_post), o_err LDPC_DBG_ARG(opt_lat_debug_));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  if (*o_err != 0) {
    // the workgroups did not come together within the bounded spins (another process holds CUs, or the device
    // is not what the occupancy query promised): do not pay that timeout on every call -- this handle decodes
    // its small batches with the batched kernels from now on
    opt_latency_ = 0;
    opt_latency_edge_ = 0;
    std::fprintf(stderr, "ldpc_toolbox (hip): the single-launch small-batch path could not get its %u workgroups resident; "
                         "this decoder uses the batched kernels from now on\n", grid);
    return kLatencyRetry;
  }
  if (host_pointers) {
    if (bits_bytes) std::memcpy(bits, d_bits, bits_bytes);
    if (iterations) std::memcpy(iterations, d_iters, batch * sizeof(int32_t));
    if (posterior) std::memcpy(posterior, d_post, post_bytes);
  }
  return 0;
}

// Largest batch the lane-per-edge path takes: 8 XCDs x the bundle an XCD decodes at once -- as many codewords as keep the
// bundle's state (soft values, messages, channel LLRs) within a few L2s' worth (measured, profiles/r03_latency.txt: 5G NR
// BG1 Zc=384 f32, 0.6 MB per codeword: ahead of the batched kernels up to 64; DVB-S2 1/2 Phif64, 3 MB: up to 32); the
// A-Min* rule's serial fold is repeated by every lane of a row: half of that.
size_t DeviceDecoder::edge_latency_limit() const {
  if (!lat_edge_ || opt_latency_edge_ == 0) return 0;
  const size_t elem = impl_.f64 ? 8 : 4;
  const size_t state = (n_ * (impl_.schedule == Schedule::Layered ? 1 : 2) + edge_lanes_) * elem;
  size_t bundle = std::max<size_t>(1, std::min<size_t>(8, (size_t(12) << 20) / std::max<size_t>(state, 1)));
  if (impl_.rule == Rule::Aminstar) bundle = std::max<size_t>(1, bundle / 2);
  // More codewords than 8 XCDs x bundle take further rounds inside the same launch.  A round costs what the first one
  // did while the batched kernels' time hardly grows with the batch at these sizes, so one extra round is where it ends:
  // BG1 Zc=384 HLTanhf32 128 / 192 / 256 codewords 2.9 / 4.4 / 6.1 ms in two / three / four rounds against 3.4 / 3.9 /
  // 4.7 ms batched, HLMinstarapproxi8 2.7 / 3.9 / 5.4 against 3.3 / 3.5 / 3.7 (profiles/r04_latency.txt; round 3 allowed
  // four rounds on the strength of a batched column timed on a cold chip).  Layered min-sum ties at one round; the
  // flooding schedule on a long code is level with the batched kernels from about 32 codewords (DVB-S2 1/2 Tanhf32:
  // 33 / 64 codewords 3.4 / 5.6 ms against 2.9 / 3.1): half the bundle there.
  if (impl_.schedule == Schedule::Flooding && n_ >= 16384) bundle = std::max<size_t>(1, std::min<size_t>(bundle, 4));
  const size_t rounds = (impl_.schedule == Schedule::Layered && impl_.rule != Rule::Minsum) ? 2 : 1;
  return std::min<size_t>(opt_latency_edge_, 8 * bundle * rounds);
}

// the lane-per-edge path (latency_edge.hip.h): layered schedule, and flooding for everything but Minsumf32
namespace {
template <int RULE, typename T, typename SrcT>
const void *edge_kernel_s(bool layered) {
  return layered ? reinterpret_cast<const void *>(dev::latency_edge_kernel<RULE, T, SrcT, true>)
                 : reinterpret_cast<const void *>(dev::latency_edge_kernel<RULE, T, SrcT, false>);
}
template <typename T, typename SrcT>
const void *edge_kernel_r(Rule rule, bool layered) {
  switch (rule) {
    case Rule::Phi: return edge_kernel_s<dev::kRulePhi, T, SrcT>(layered);
    case Rule::Tanh: return edge_kernel_s<dev::kRuleTanh, T, SrcT>(layered);
    case Rule::Minstarapprox: return edge_kernel_s<dev::kRuleMinstarapprox, T, SrcT>(layered);
    case Rule::Aminstar: return edge_kernel_s<dev::kRuleAminstar, T, SrcT>(layered);
    default: return edge_kernel_s<dev::kRuleMinsum, T, SrcT>(layered);
  }
}
const void *edge_kernel(Rule rule, bool arith_i8, bool arith_f64, bool src_f64, bool layered) {
  if (arith_i8)  // the rule (Minstarapprox / A-Min*) and its options are run-time arguments (dev::I8Opts)
    return src_f64 ? edge_kernel_s<dev::kRuleEdgeI8, int32_t, double>(layered) : edge_kernel_s<dev::kRuleEdgeI8, int32_t, float>(layered);
  if (arith_f64) return src_f64 ? edge_kernel_r<double, double>(rule, layered) : edge_kernel_r<double, float>(rule, layered);
  return src_f64 ? edge_kernel_r<float, double>(rule, layered) : edge_kernel_r<float, float>(rule, layered);
}
}  // namespace

int DeviceDecoder::decode_latency_edge(const void *llrs, bool llrs_f64, bool host_pointers, size_t batch,
                                          uint32_t max_iterations, uint8_t *bits, size_t out_len, int32_t *iterations,
                                          void *posterior, hipStream_t s) {
  std::lock_guard<std::mutex> one_at_a_time(g_latency_mutex);
  EdgeLatencyPath &lp = *lat_edge_;
  const size_t in_elem = llrs_f64 ? 8 : 4, elem = impl_.f64 ? 8 : 4;
  const uint32_t n = static_cast<uint32_t>(n_), m = static_cast<uint32_t>(m_);
  last_lanes_ = 1;
  last_group_ = batch;
  if (!lp.uploaded) {
    auto up = [&](const std::vector<uint32_t> &v, uint32_t **dst) -> int {
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(v.size(), 1) * sizeof(uint32_t)));
      if (!v.empty()) HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      return 0;
    };
    if (int rc = up(lp.h_level_chunk, &lp.d_level_chunk)) return rc;
    if (int rc = up(lp.h_lane_var, &lp.d_lane_var)) return rc;
    if (int rc = up(lp.h_lane_info, &lp.d_lane_info)) return rc;
    if (int rc = up(lp.h_var_ptr, &lp.d_var_ptr)) return rc;
    if (int rc = up(lp.h_var_lane, &lp.d_var_lane)) return rc;
    // per-XCD codeword state, each array on a 256-byte boundary: soft values | messages (one per lane slot) |
    // channel LLRs (flooding) | raw hard decisions
    const size_t a_q = round_up(size_t(n) * elem + 256, 256), a_r = round_up(size_t(lp.n_chunks) * 64 * elem + 256, 256),
                 a_c = lp.layered ? 0 : a_q, a_h = round_up(size_t(n) + 256, 256), slot = a_q + a_r + a_c + a_h;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lp.slots.base), size_t(8) * dev::kEdgeBundle * slot));
    lp.slots.slot_bytes = slot;
    lp.slots.off_msg = a_q;
    lp.slots.off_chan = a_q + a_r;
    lp.slots.off_rawhard = a_q + a_r + a_c;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lp.slots.flags), size_t(8) * 2 * dev::kEdgeBundle * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lp.d_sync), sizeof(dev::LatencySync)));
    lp.uploaded = true;
  }
  if (lp.grid == 0) {
    // every workgroup of the persistent launch must be resident (see decode_latency)
    int cus = 0, per_cu_f = 0, per_cu_d = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_f, edge_kernel(impl_.rule, impl_.i8, impl_.f64, false, lp.layered), 1024, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_d, edge_kernel(impl_.rule, impl_.i8, impl_.f64, true, lp.layered), 1024, 0);
    const int resident = e == hipSuccess ? cus * std::min(per_cu_f, per_cu_d) : 0;
    if (resident < 8) {
      opt_latency_ = 0;
      opt_latency_edge_ = 0;
      return kLatencyRetry;
    }
    lp.grid = static_cast<uint32_t>(std::min<int>(resident, opt_lat_grid_ ? static_cast<int>(opt_lat_grid_) : 256));
  }
  const void *d_llrs = llrs;
  uint8_t *d_bits = bits;
  int32_t *d_iters = iterations;
  void *d_post = posterior;
  const size_t in_bytes = batch * input_len_ * in_elem, bits_bytes = batch * out_len, post_bytes = batch * n_ * in_elem;
  const size_t iters_at = round_up(256 + bits_bytes, 256), post_at = round_up(iters_at + batch * sizeof(int32_t), 256);
  const size_t out_need = host_pointers ? post_at + (posterior ? post_bytes : 0) : 256;
  if (EdgeLatencyPath::pinned(&lp.h_out, &lp.h_out_bytes, out_need) ||
      (host_pointers && EdgeLatencyPath::pinned(&lp.h_in, &lp.h_in_bytes, in_bytes))) {
    fail("pinned host memory for the small-batch path");
    return -1;
  }
  uint32_t *o_err = reinterpret_cast<uint32_t *>(lp.h_out);
  *o_err = 0;
  if (host_pointers) {
    std::memcpy(lp.h_in, llrs, in_bytes);
    d_llrs = lp.h_in;
    d_bits = reinterpret_cast<uint8_t *>(lp.h_out + 256);
    d_iters = reinterpret_cast<int32_t *>(lp.h_out + iters_at);
    d_post = posterior ? static_cast<void *>(lp.h_out + post_at) : nullptr;
  }
  HIP_TRY(hipMemsetAsync(lp.d_sync, 0, sizeof(dev::LatencySync), s));
  // up to 8 codewords: one per XCD; more: every XCD takes a bundle of up to kEdgeBundle that share each phase and barrier
  uint32_t bundle = static_cast<uint32_t>(std::min<size_t>(dev::kEdgeBundle, (batch + 7) / 8));
  if (bundle > 1) HIP_TRY(hipMemsetAsync(lp.slots.flags, 0, size_t(8) * 2 * dev::kEdgeBundle * sizeof(uint32_t), s));
  dev::EdgeLatTables t{n, m, static_cast<uint32_t>(lp.h_level_chunk.size() - 1), lp.n_chunks, lp.d_level_chunk, lp.d_lane_var,
                       lp.d_lane_info, lp.d_var_ptr, lp.d_var_lane, d_src_block_, pattern_len_ ? n / pattern_len_ : 0};
  uint32_t in_len = static_cast<uint32_t>(input_len_), nb = static_cast<uint32_t>(batch), ol = static_cast<uint32_t>(out_len);
  dev::I8Opts i8o{impl_.rule == Rule::Aminstar, impl_.jones, impl_.hardlimit, impl_.deg1clip};
  void *args[] = {&t, &lp.slots, &lp.d_sync, &d_llrs, &in_len, &nb, &max_iterations, &d_bits, &ol, &d_iters, &d_post, &o_err, &bundle, &i8o};
  HIP_TRY(hipLaunchKernel(edge_kernel(impl_.rule, impl_.i8, impl_.f64, llrs_f64, lp.layered), dim3(lp.grid), dim3(1024), args, 0, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (*o_err != 0) {
    opt_latency_ = 0;  // see decode_latency
    opt_latency_edge_ = 0;
    std::fprintf(stderr, "ldpc_toolbox (hip): the single-launch small-batch path could not get its %u workgroups resident; "
                         "this decoder uses the batched kernels from now on\n", lp.grid);
    return kLatencyRetry;
  }
  if (host_pointers) {
    if (bits_bytes) std::memcpy(bits, d_bits, bits_bytes);
    if (iterations) std::memcpy(iterations, d_iters, batch * sizeof(int32_t));
    if (posterior) std::memcpy(posterior, d_post, post_bytes);
  }
  return 0;
}

// ---- continuous batching -----------------------------------------------------------------------
// (exact -- same counters as drained batches and the CPU checker -- and slower in this layout: 0.64-0.69 of the
// iteration-proportional bound against 0.75-0.81, profiles/r03_continuous_batching.txt.  Since round 5 only builds with
// -DLDPC_EXPERIMENTS carry it; in the product stream_capable() is false and the simulator's "streaming" option changes nothing.)
bool DeviceDecoder::stream_capable() const {
#ifdef LDPC_EXPERIMENTS
  return impl_.schedule == Schedule::Flooding && impl_.rule == Rule::Minsum && !impl_.f64 && !impl_.i8 && rec_ready_ &&
         lfree_ready_ && records_wanted() && opt_lfree_ && !opt_staged_minsum_;
#else
  return false;
#endif
}

#ifndef LDPC_EXPERIMENTS
int DeviceDecoder::decode_stream(const std::function<void(const uint64_t *, float *, hipStream_t)> &, float *, size_t total,
                                 uint32_t, uint8_t *, size_t, int32_t *) {
  if (total == 0) return 0;
  fail("decode_stream: continuous batching is an experiment build's feature (-DLDPC_EXPERIMENTS)");
  return -3;
}
#else

int DeviceDecoder::decode_stream(const std::function<void(const uint64_t *, float *, hipStream_t)> &source, float *staging,
                                 size_t total, uint32_t max_iterations, uint8_t *bits, size_t out_len, int32_t *iterations) {
  if (total == 0) return 0;
  if (!stream_capable() || max_iterations == 0 || total >= (size_t(1) << 32)) {
    fail("decode_stream: flooding Minsumf32 with row records only, at least one iteration");
    return -3;
  }
  if (out_len > n_) {
    fail("output_len larger than the codeword length");
    return -1;
  }
  typedef float T;
  HIP_TRY(hipSetDevice(device_));
  const size_t G = stream_group();
  Workspace &w = *ws_[0];
  if (int rc = ensure_lanes(1, G)) return rc;
  hipStream_t s = stream_;
  if (int rc = order_after_default_stream(s)) return rc;
  const uint32_t n = static_cast<uint32_t>(n_), m = static_cast<uint32_t>(m_), Gu = static_cast<uint32_t>(G), W = Gu / 64;
  uint32_t tile = opt_tile_ ? opt_tile_ : 256;
  tile = std::max<uint32_t>(64, tile / 64 * 64);
  while (Gu % tile != 0) tile -= 64;
  const uint32_t vec = std::max<uint32_t>(2, pick_vec_for(tile, 4, opt_vec_));
  const bool fits32 = uint64_t(std::max<size_t>(std::max(e_, n_), m_ * rec_w_)) * tile * sizeof(T) < (1ull << 32);
  if (!w.records || !w.rec[0] || !w.d_flag || !fits32 || tile % (64 * vec) != 0) {
    fail("decode_stream: the row-record workspace is not available for this graph");
    return -3;
  }
  T *chan = static_cast<T *>(w.chan), *post = static_cast<T *>(w.post), *msg = static_cast<T *>(w.msg);
  T *rbuf[2] = {static_cast<T *>(w.rec[0]), static_cast<T *>(w.rec[1])};
  g_knobs.nt_vn = opt_nt_vn_;
  g_knobs.row_scratch = nullptr;
  dev::Graph g{d_row_ptr_, d_edge_col_, d_col_ptr_, d_col_edge_, m, n, static_cast<uint32_t>(e_),
               nullptr,    nullptr,     nullptr,    0,           d_edge_aux_, d_edge_peer_};
  dev::Graph g_keep = g;
  g_keep.list_var = d_keep_var_;
  g_keep.list_ptr = d_keep_ptr_;
  g_keep.list_edge = d_keep_pos_;
  g_keep.n_list = n_keep_;
  dev::State st{w.done, w.iters, w.n_active, w.n_slots, w.slot_cw, nullptr, 0, 0, nullptr, w.it0, max_iterations};
  w.epoch = (w.epoch % 0xFFFFFFu) + 1;
  const uint32_t stream_block = 256, target_waves = opt_waves_ ? opt_waves_ : 256 * 1024;
  const uint32_t rec_run = std::max<uint32_t>(1, std::min<uint32_t>(opt_rec_run_, m));
  const Tiling rec_t = make_tiling(Gu, tile, 64 * vec, (m + rec_run - 1) / rec_run, stream_block, target_waves);
  const Tiling vn_keep_t = make_tiling(Gu, tile, 64 * vec, n_keep_, stream_block, opt_waves_vn_ ? opt_waves_vn_ : 128 * 1024);

  // every slot starts empty: finished, no codeword
  dev::init_group_kernel<<<(Gu + 255) / 256, 256, 0, s>>>(w.done, w.iters, w.unsat0, w.unsat1, w.n_active, w.n_slots, w.slot_cw,
                                                         0u, Gu);
  HIP_TRY(hipMemsetAsync(w.it0, 0, G * sizeof(uint32_t), s));
  dev::StreamPlan plan0{};
  plan0.total = total;
  HIP_TRY(hipMemcpyAsync(w.stream_plan, &plan0, sizeof(plan0), hipMemcpyHostToDevice, s));
  // n_slots = the whole group for the whole call (init_group_kernel sized it for zero codewords)
  const uint32_t all_slots = Gu;
  HIP_TRY(hipMemcpyAsync(w.n_slots, &all_slots, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  *w.h_flag = 0;
  const uint32_t block_size = pattern_len_ ? n / pattern_len_ : 0;
  auto harvest = [&](uint32_t now) {
    // results of the finished codewords -> the caller's rows; their slots (and the never-filled ones) -> the next codewords
    dim3 egrid(std::min<uint32_t>((static_cast<uint32_t>(std::max<size_t>(out_len, 1)) + 63) / 64, 1024), W);
    dev::emit_kernel<T, float><<<egrid, 256, 0, s>>>(post, nullptr, st, &w.stream_plan->always, n, Gu, tile,
                                                    static_cast<uint32_t>(out_len), bits, iterations, nullptr, 0, 1);
    dev::stream_plan_kernel<<<1, 1024, 0, s>>>(st, w.stream_plan, w.holes, Gu, w.d_flag, w.epoch);
    source(reinterpret_cast<const uint64_t *>(w.stream_plan), staging, s);
    dim3 igrid((n + 63) / 64, W);
    dev::stream_ingest_kernel<float, T><<<igrid, 256, 0, s>>>(staging, input_len_, w.stream_plan, w.holes, st, w.it0, now, n, tile,
                                                             chan, post, w.unsat0, w.unsat1, d_src_block_, block_size);
  };
  auto retired = [&]() -> uint64_t {
    const uint64_t f = ProgressPoll::load(w.h_flag);
    return (f >> 40) == uint64_t(w.epoch & 0xFFFFFFu) ? (f & 0xFFFFFFFFFFull) : 0;
  };
  for (auto &e : stream_events_)
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  harvest(0);
  uint32_t *unsat[2] = {w.unsat0, w.unsat1};
  const uint32_t every = std::max<uint32_t>(1, opt_stream_harvest_);
  // the host enqueues ahead of the device; it stops when the device has reported the last codeword retired, and
  // never runs more than a few harvests ahead of what the device has reported (the launches after the end would
  // all return at once, but there is no point in queueing thousands of them)
  const uint64_t upper = (uint64_t(total) / G + 2) * (uint64_t(max_iterations) + every + 1) + 8;  // cannot take longer
  for (uint64_t it = 1; it <= upper; it++) {
    last_stream_iterations_ = it;
    dev::State stp = st;
    stp.tick = static_cast<uint32_t>(it);
    uint32_t *unsat_out = unsat[it & 1];
    Launch<T>::cn_rec_stream(vec, rec_w_, rec_t, s, g, stp, chan, post, rbuf[(it + 1) & 1], rbuf[it & 1], msg, unsat_out, rec_run);
    Launch<T>::vn(true, vec, opt_unroll_vn_, vn_keep_t, s, g_keep, st, chan, msg, post, unsat_out, unsat[(it + 1) & 1],
                  static_cast<int32_t>(it) - 1);
    if (it % every == 0) {
      harvest(static_cast<uint32_t>(it));
      if (retired() >= total) break;
      // never more than kAhead harvests ahead of the device (an event per harvest, waited for kAhead harvests later)
      const uint64_t h = it / every;
      HIP_TRY(hipEventRecord(stream_events_[h % kStreamEvents], s));
      if (h >= kStreamAhead) {
        HIP_TRY(hipEventSynchronize(stream_events_[(h - kStreamAhead) % kStreamEvents]));
        if (retired() >= total) break;
      }
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (retired() < total) {
    fail("decode_stream: the stream did not drain (internal error)");
    return -2;
  }
  return 0;
}
#endif  // LDPC_EXPERIMENTS (continuous batching)

// ---- syndrome operator ----------------------------------------------------------------------

int DeviceDecoder::syndrome_device(const uint8_t *bits, size_t batch, uint8_t *syndrome, uint32_t *weight,
                                   hipStream_t stream) {
  if (batch == 0 || (!syndrome && !weight)) return 0;
  if (batch > 65535) {
    fail("syndrome: more than 65535 codewords in one call");
    return -1;
  }
  HIP_TRY(hipSetDevice(device_));
  const bool own_stream = stream == nullptr;
  hipStream_t s = own_stream ? stream_ : stream;
  if (own_stream)
    if (int rc = order_after_default_stream(s)) return rc;
  if (weight) HIP_TRY(hipMemsetAsync(weight, 0, batch * sizeof(uint32_t), s));
  const uint32_t m = static_cast<uint32_t>(m_);
  dim3 grid(std::max<uint32_t>((m + 255) / 256, 1), static_cast<uint32_t>(batch));
  dev::syndrome_of_bits_kernel<<<grid, 256, 0, s>>>(d_row_ptr_, d_edge_col_, m, static_cast<uint32_t>(n_),
                                                    static_cast<uint32_t>(batch), bits, syndrome, weight);
  HIP_TRY(hipGetLastError());
  if (own_stream) HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int DeviceDecoder::syndrome_host(const uint8_t *bits, size_t batch, uint8_t *syndrome, uint32_t *weight) {
  if (batch == 0 || (!syndrome && !weight)) return 0;
  HIP_TRY(hipSetDevice(device_));
  const size_t chunk = 4096;
  uint8_t *d_bits = nullptr, *d_syn = nullptr;
  uint32_t *d_w = nullptr;
  auto release = [&]() {
    for (void *p : {(void *)d_bits, (void *)d_syn, (void *)d_w})
      if (p) (void)hipFree(p);
  };
  const size_t cap = std::min(batch, chunk);
  bool ok = hipMalloc(reinterpret_cast<void **>(&d_bits), cap * n_) == hipSuccess;
  if (ok && syndrome) ok = hipMalloc(reinterpret_cast<void **>(&d_syn), std::max<size_t>(cap * m_, 1)) == hipSuccess;
  if (ok && weight) ok = hipMalloc(reinterpret_cast<void **>(&d_w), cap * sizeof(uint32_t)) == hipSuccess;
  if (!ok) {
    release();
    fail("syndrome: device staging allocation failed");
    return -2;
  }
  int rc = 0;
  for (size_t b0 = 0; b0 < batch && rc == 0; b0 += chunk) {
    const size_t nb = std::min(chunk, batch - b0);
    if (hipMemcpyAsync(d_bits, bits + b0 * n_, nb * n_, hipMemcpyHostToDevice, stream_) != hipSuccess) rc = -2;
    if (rc == 0) rc = syndrome_device(d_bits, nb, d_syn, d_w, stream_);
    if (rc == 0 && syndrome && m_ &&
        hipMemcpyAsync(syndrome + b0 * m_, d_syn, nb * m_, hipMemcpyDeviceToHost, stream_) != hipSuccess)
      rc = -2;
    if (rc == 0 && weight &&
        hipMemcpyAsync(weight + b0, d_w, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_) != hipSuccess)
      rc = -2;
    if (rc == 0 && hipStreamSynchronize(stream_) != hipSuccess) rc = -2;
  }
  release();
  if (rc == -2 && error_.empty()) fail("syndrome: copy failed");
  return rc;
}

}  // namespace ldpc
