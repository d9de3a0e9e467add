// Host orchestration of the batched HIP decoder: graph upload, workspace in HBM,
// the per-group launch sequence of both schedules.  See device_decoder.h / DESIGN.md.
#define LDPC_GROUP_KERNELS_TU 1  // this translation unit compiles the non-template group kernels
#include "device_decoder_internal.h"

namespace ldpc {

bool DeviceDecoder::fail(const std::string &msg, hipError_t e) {
  static std::mutex m;  // (the execution lanes' enqueuing threads may both fail)
  std::lock_guard<std::mutex> lock(m);
  error_ = msg;
  if (e != hipSuccess) error_ += std::string(": ") + hipGetErrorString(e);
  std::fprintf(stderr, "ldpc_toolbox (hip): %s\n", error_.c_str());
  return false;
}

// the launchers declared in device_decoder_internal.h
namespace grp {
void init_group(hipStream_t s, uint32_t *done, int32_t *iters, uint32_t *unsat0, uint32_t *unsat1, uint32_t *n_active,
                uint32_t *n_slots, uint32_t *slot_cw, uint32_t nb, uint32_t G) {
  dev::init_group_kernel<<<(G + 255) / 256, 256, 0, s>>>(done, iters, unsat0, unsat1, n_active, n_slots, slot_cw, nb, G);
}
void latch(hipStream_t s, uint32_t *done, int32_t *iters, uint32_t *unsat, uint32_t *n_active, int32_t iteration, uint32_t G) {
  dev::latch_kernel<<<(G + 255) / 256, 256, 0, s>>>(done, iters, unsat, n_active, iteration, G);
}
void syndrome_bits(hipStream_t s, uint32_t threads, const uint32_t *row_ptr, const uint32_t *edge_col, uint32_t n_rows,
                   const uint64_t *bits, uint32_t *unsat, const uint32_t *n_active, const uint32_t *n_slots, uint32_t W,
                   uint32_t rows_per_thread) {
  dev::syndrome_bits_kernel<<<(threads + 255) / 256, 256, 0, s>>>(row_ptr, edge_col, n_rows, bits, unsat, n_active, n_slots, W,
                                                                  rows_per_thread);
}
void compact_plan(hipStream_t s, dev::State st, dev::CompactPlan *plan, uint32_t *movers, uint32_t *holes, uint32_t *fill_cw,
                  uint32_t remaining_iterations, dev::CompactRule rule) {
  dev::compact_plan_kernel<<<1, 1024, 0, s>>>(st, plan, movers, holes, fill_cw, remaining_iterations, rule);
}
void compact_commit(hipStream_t s, dev::State st, const dev::CompactPlan *plan, uint32_t *unsat0, uint32_t *unsat1,
                    uint32_t *n_slots, const uint32_t *fill_cw, uint32_t G) {
  dev::compact_commit_kernel<<<(G + 255) / 256, 256, 0, s>>>(st, plan, unsat0, unsat1, n_slots, fill_cw, G);
}
}  // namespace grp

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
  d->opt_vec_ = env_u32("LDPC_TOOLBOX_VEC", 4);
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
  if (pool_) {
    pool_->release();
    delete pool_;
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
  else if (key == "vec")
    opt_vec_ = v;
  else if (key == "tile")
    opt_tile_ = v;
  else if (key == "lfree")
    opt_lfree_ = v != 0;
  else if (key == "records")
    opt_records_ = v != 0 ? (v >= 2 ? 2 : 1) : 0;  // 2: also where the graph's peers are distant rows
  else if (key == "rec_run")
    opt_rec_run_ = std::max<uint32_t>(v, 1);
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
  else if (key == "rec_long")
    opt_rec_long_ = v != 0;
  else if (key == "compact")
    opt_compact_ = v != 0;
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
  else if (key == "lane_threads")
    opt_lane_threads_ = v != 0;
  else if (key == "throttle")
    opt_throttle_ = v != 0;
  else if (key == "pooling")
    opt_pooling_ = v != 0;
  else if (key == "lead")
    opt_lead_ = v;
  else if (key == "lane_pace")
    opt_lane_pace_ = v != 0;
  else if (key == "lanes")
    opt_lanes_ = std::min<uint32_t>(v, 2);
  else if (key == "poll")
    opt_poll_ = v != 0;
  else if (key == "latency") {
    opt_latency_ = v;
    opt_latency_edge_ = v == 0 ? 0 : std::max<uint32_t>(v, 64);  // 0 switches both small-batch paths off
  } else if (key == "latency_edge")
    opt_latency_edge_ = v;
  else if (key == "compact_first")
    opt_compact_first_ = v;
  else if (key == "serial_levels")
    opt_serial_levels_ = v;
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
    if (w.G == G && w.elem == elem && w.chan && w.records == records && (!place || w.slab == place))
      return 0;
    w.release();
  }
  const size_t W = G / 64;
  // One slab, carved: the big arrays first, each start 2 MiB-aligned.
  // (Separate hipMalloc calls made the check-node kernel's time vary by ~12 % from one
  // allocation to the next; a single slab keeps the relative placement fixed.)
  const size_t align = size_t(2) << 20;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t at = off;
    off = round_up(off + std::max<size_t>(bytes, 256), align);
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
  const size_t lane_align = size_t(2) << 20;
  const size_t stride = round_up(bytes, lane_align);
  const size_t elem = impl_.i8 ? 2 : (impl_.f64 ? 8 : 4);
  const bool records = lfree_ready_ && rec_ready_ && records_wanted() && opt_lfree_;
  auto current = [&](const Workspace &w, const char *at) {
    return w.borrowed && w.slab == at && w.G == G && w.elem == elem && w.records == records;
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

// ---- straggler pooling inside the batch entries (device_decoder.h, "pooling") -------------------------------------------

int DeviceDecoder::ensure_pool(size_t batch, size_t rows, size_t out_len, size_t in_elem, bool posterior, bool own_iterations) {
  if (!pool_) pool_ = new StragglerPool();
  StragglerPool &p = *pool_;
  auto grow = [&](void **ptr, size_t *have, size_t need) -> int {
    if (*have >= need) return 0;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(ptr, std::max<size_t>(need, 256)));
    *have = need;
    return 0;
  };
  if (int rc = grow(reinterpret_cast<void **>(&p.d_idx), &p.idx_cap, batch * sizeof(uint32_t))) return rc;
  if (!p.d_stats) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p.d_stats), 64));
  if (int rc = grow(&p.d_llrs, &p.llr_bytes, rows * input_len_ * in_elem)) return rc;
  if (int rc = grow(reinterpret_cast<void **>(&p.d_bits), &p.bits_bytes, rows * std::max<size_t>(out_len, 1))) return rc;
  if (int rc = grow(reinterpret_cast<void **>(&p.d_its), &p.its_rows, rows * sizeof(int32_t))) return rc;
  if (posterior)
    if (int rc = grow(&p.d_post, &p.post_bytes, rows * n_ * in_elem)) return rc;
  if (own_iterations)
    if (int rc = grow(reinterpret_cast<void **>(&p.d_its_all), &p.its_all, batch * sizeof(int32_t))) return rc;
  return 0;
}

// The next chunk's budget from what the call has seen (the simulation driver's rule, csrc/simulator.hip): twice the average
// iteration count of the frames that converge, plus 8 -- a straggler counts with the budget it has exhausted (a lower bound;
// left out, the average would cover only the frames that beat the budget and ratchet it down) -- and the full budget again
// when more than a quarter of the frames have needed (or would need) the second pass, or when there is little to gain.
uint32_t DeviceDecoder::next_pool_budget(double ok, double sum_its_ok, double stragglers, double straggler_its,
                                         double failed_full, uint32_t max_it) {
  const double ok_frames = ok + stragglers, ok_its = sum_its_ok + straggler_its;
  const double avg_ok = ok_frames > 0 ? ok_its / ok_frames : static_cast<double>(max_it);
  uint32_t next = static_cast<uint32_t>(std::min<double>(max_it, std::ceil(2.0 * avg_ok) + 8.0));
  next = std::max<uint32_t>(next, 16);
  if (stragglers + failed_full > 0.25 * (ok + stragglers + failed_full) || uint64_t(next) * 10 >= uint64_t(max_it) * 7) next = max_it;
  return next;
}

// frames per chunk of a pooled call: what the execution lanes decode at once
size_t DeviceDecoder::pool_chunk(size_t batch) const { return pick_group(batch) * std::max<uint32_t>(lane_count(), 1); }

int DeviceDecoder::decode_device(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits,
                                 size_t out_len, int32_t *iterations, void *posterior, hipStream_t stream) {
  last_pooled_ = 0;
  // (a call on the CALLER's stream only enqueues -- unless "throttle" lets it wait: pooling needs every chunk's counts back)
  if (opt_pooling_ && max_iterations >= 24 && batch >= 2 * pool_chunk(batch) && (stream == nullptr || opt_throttle_) && out_len <= n_)
    return decode_device_pooled(llrs, llrs_f64, batch, max_iterations, bits, out_len, iterations, posterior, stream);
  return decode_device_plain(llrs, llrs_f64, batch, max_iterations, bits, out_len, iterations, posterior, stream);
}

int DeviceDecoder::decode_device_pooled(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits,
                                        size_t out_len, int32_t *iterations, void *posterior, hipStream_t stream) {
  HIP_TRY(hipSetDevice(device_));
  const size_t in_elem = llrs_f64 ? 8 : 4, G = pick_group(batch), chunk = pool_chunk(batch);
  hipStream_t s = stream ? stream : stream_;
  if (int rc = ensure_pool(batch, chunk, out_len, in_elem, posterior != nullptr, iterations == nullptr)) return rc;
  StragglerPool &p = *pool_;
  int32_t *its = iterations ? iterations : p.d_its_all;
  unsigned long long *d_sum = reinterpret_cast<unsigned long long *>(p.d_stats + 4);
  HIP_TRY(hipMemsetAsync(p.d_stats, 0, 64, s));
  uint32_t budget = (pool_budget_ && pool_budget_max_it_ == max_iterations) ? std::min(pool_budget_, max_iterations) : max_iterations;
  double straggler_its = 0.0;
  uint32_t seen_stragglers = 0;
  const size_t llr_row = input_len_ * in_elem, post_row = n_ * in_elem;
  for (size_t b0 = 0; b0 < batch; b0 += chunk) {
    const size_t nf = std::min(chunk, batch - b0);
    if (int rc = decode_device_plain(static_cast<const char *>(llrs) + b0 * llr_row, llrs_f64, nf, budget, bits + b0 * out_len, out_len,
                                     its + b0, posterior ? static_cast<char *>(posterior) + b0 * post_row : nullptr, stream))
      return rc;
    dev::pool_select_kernel<<<static_cast<uint32_t>((nf + 255) / 256), 256, 0, s>>>(
        its + b0, static_cast<uint32_t>(nf), static_cast<uint32_t>(b0), budget < max_iterations ? 1 : 0, p.d_idx, p.d_stats, d_sum);
    uint32_t h[6];
    HIP_TRY(hipMemcpyAsync(h, p.d_stats, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    unsigned long long sum_ok;
    std::memcpy(&sum_ok, &h[4], sizeof(sum_ok));
    straggler_its += double(h[2] - seen_stragglers) * budget;
    seen_stragglers = h[2];
    budget = next_pool_budget(h[0], static_cast<double>(sum_ok), h[2], straggler_its, h[1], max_iterations);
  }
  pool_budget_ = budget;
  pool_budget_max_it_ = max_iterations;
  // the stragglers, together, with the full budget (a handful takes the single-launch small-batch path; more run in the
  // chunks' group size, so that the decoder keeps the workspace it has)
  const size_t keep_min = min_group_;
  for (size_t p0 = 0; p0 < seen_stragglers; p0 += chunk) {
    const uint32_t np = static_cast<uint32_t>(std::min<size_t>(chunk, seen_stragglers - p0));
    const uint32_t blocks = (np * 64 + 255) / 256;
    dev::pool_gather_kernel<<<blocks, 256, 0, s>>>(p.d_idx + p0, np, static_cast<const uint32_t *>(llrs), llr_row / 4,
                                                   static_cast<uint32_t *>(p.d_llrs));
    min_group_ = G;
    const int rc = decode_device_plain(p.d_llrs, llrs_f64, np, max_iterations, p.d_bits, out_len, p.d_its, posterior ? p.d_post : nullptr, stream);
    min_group_ = keep_min;
    if (rc) return rc;
    dev::pool_scatter_kernel<<<blocks, 256, 0, s>>>(p.d_idx + p0, np, p.d_bits, static_cast<uint32_t>(out_len), p.d_its,
                                                    static_cast<const uint32_t *>(p.d_post), posterior ? post_row / 4 : 0, bits,
                                                    iterations, static_cast<uint32_t *>(posterior));
  }
  last_pooled_ = seen_stragglers;
  HIP_TRY(hipGetLastError());
  if (stream == nullptr) HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int DeviceDecoder::decode_device_plain(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations,
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
      if (int rc = run_any(*ws_[lane], src, llrs_f64, nb, max_iterations, dst_bits, out_len, dst_it, dst_post,
                           lane ? stream2_ : s, may_block, threaded))
        return rc;
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

int DeviceDecoder::decode_host(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits,
                               size_t out_len, int32_t *iterations, void *posterior) {
  last_pooled_ = 0;
  if (opt_pooling_ && max_iterations >= 24 && batch >= 2 * 4 * pool_chunk(batch) && out_len <= n_)
    return decode_host_pooled(llrs, llrs_f64, batch, max_iterations, bits, out_len, iterations, posterior);
  return decode_host_plain(llrs, llrs_f64, batch, max_iterations, bits, out_len, iterations, posterior);
}

// Host buffers: the caller's rows stay where they are, so the pool is a list of frame indices; chunks of four times what
// the lanes decode at once (a chunk is one pipelined call of the plain entry: its first copy in and last copy out are exposed).
int DeviceDecoder::decode_host_pooled(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations, uint8_t *bits,
                                      size_t out_len, int32_t *iterations, void *posterior) {
  const size_t in_elem = llrs_f64 ? 8 : 4, chunk = 4 * pool_chunk(batch);
  const size_t llr_row = input_len_ * in_elem, post_row = n_ * in_elem;
  std::vector<int32_t> own_its;
  if (!iterations) own_its.resize(batch);
  int32_t *its = iterations ? iterations : own_its.data();
  std::vector<size_t> stragglers;
  uint32_t budget = (pool_budget_ && pool_budget_max_it_ == max_iterations) ? std::min(pool_budget_, max_iterations) : max_iterations;
  double ok = 0, sum_ok = 0, straggler_its = 0, failed_full = 0;
  for (size_t b0 = 0; b0 < batch; b0 += chunk) {
    const size_t nf = std::min(chunk, batch - b0);
    if (int rc = decode_host_plain(static_cast<const char *>(llrs) + b0 * llr_row, llrs_f64, nf, budget, bits + b0 * out_len, out_len,
                                   its + b0, posterior ? static_cast<char *>(posterior) + b0 * post_row : nullptr))
      return rc;
    const bool reduced = budget < max_iterations;
    for (size_t i = 0; i < nf; i++) {
      const int32_t it = its[b0 + i];
      if (it >= 0) {
        ok += 1;
        sum_ok += it;
      } else if (reduced) {
        stragglers.push_back(b0 + i);
        straggler_its += budget;
      } else {
        failed_full += 1;
      }
    }
    budget = next_pool_budget(ok, sum_ok, static_cast<double>(stragglers.size()), straggler_its, failed_full, max_iterations);
  }
  pool_budget_ = budget;
  pool_budget_max_it_ = max_iterations;
  if (!stragglers.empty()) {
    const size_t np = stragglers.size();
    std::vector<char> rows(np * llr_row), pbits(np * std::max<size_t>(out_len, 1)), ppost(posterior ? np * post_row : 0);
    std::vector<int32_t> pits(np);
    for (size_t i = 0; i < np; i++) std::memcpy(&rows[i * llr_row], static_cast<const char *>(llrs) + stragglers[i] * llr_row, llr_row);
    const size_t keep_min = min_group_;
    min_group_ = pick_group(batch);
    const int rc = decode_host_plain(rows.data(), llrs_f64, np, max_iterations, reinterpret_cast<uint8_t *>(pbits.data()), out_len,
                                     pits.data(), posterior ? ppost.data() : nullptr);
    min_group_ = keep_min;
    if (rc) return rc;
    for (size_t i = 0; i < np; i++) {
      const size_t f = stragglers[i];
      std::memcpy(bits + f * out_len, &pbits[i * out_len], out_len);
      its[f] = pits[i];
      if (posterior) std::memcpy(static_cast<char *>(posterior) + f * post_row, &ppost[i * post_row], post_row);
    }
  }
  last_pooled_ = stragglers.size();
  return 0;
}

int DeviceDecoder::decode_host_plain(const void *llrs, bool llrs_f64, size_t batch, uint32_t max_iterations,
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
    // decode all the same, and the FIRST copy overlaps nothing -- a quarter group's copy is a quarter of that exposure)
    size_t b0 = 0;
    if (batch >= 2 * G && G >= 1024 && (G / 4) % 256 == 0) {
      starts.push_back(0);
      starts.push_back(G / 4);
      b0 = G;
    }
    for (; b0 < batch; b0 += G) starts.push_back(b0);
    // ... and closes with a short group: the last group's results are the only ones whose way back is exposed
    const size_t last0 = starts.back(), last_n = batch - last0;
    if (starts.size() >= 3 && last_n >= 1024) {
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
#endif  // (the experiment builds' decode_stream: decode_stream.hip.h, compiled with the f32 kernels)

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
