// One group of codewords through the float rules, both schedules: DeviceDecoder::run_group<T>.  Instantiated for float in
// run_group_f32.hip and for double in run_group_f64.hip (each compiles its own kernels).
#pragma once
#include "launch.hip.h"

namespace ldpc {

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

  // layout tile: codewords per self-contained sub-batch (kernels.hip.h, tile_base)
  uint32_t tile = opt_tile_ ? opt_tile_ : (sizeof(T) == 4 ? 256 : 128);
  tile = std::max<uint32_t>(64, tile / 64 * 64);
  while (G % tile != 0) tile -= 64;

  g_knobs.rec_dbg = opt_rec_dbg_;
  g_knobs.rec_long = max_row_weight_ > 8 || opt_rec_long_;
  g_knobs.fast = impl_.fast;
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

  grp::init_group(s, w.done, w.iters, w.unsat0, w.unsat1, w.n_active, w.n_slots,
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
      std::max<uint32_t>(1, std::min<uint32_t>(64, uint32_t(uint64_t(m) * synd_chunks * 64 / kSyndThreads)));
  const uint32_t synd_threads = 64 * synd_chunks * ((m + synd_rows - 1) / synd_rows);
  auto syndrome_of = [&](const uint64_t *hard, uint32_t *unsat) {
    if (m == 0) return;
    grp::syndrome_bits(s, synd_threads, d_row_ptr_, d_edge_col_, m, hard,
                                                                         unsat, w.n_active, w.n_slots, W, synd_rows);
  };
  auto latch = [&](uint32_t *unsat, int32_t it) {
    grp::latch(s, w.done, w.iters, unsat, w.n_active, it, G);
  };
  // (one codeword per lane: the paired-load form of the 16-bit posterior, pack_hard_pair_kernel, is slower here --
  // 210 vs 191 us for 8192 x BG1 Zc=384: 256-byte requests already stream, the exchange only adds work)
  // (16 K waves by default: the launch runs once per layered iteration and its waves are short -- at the 256 K of the other
  // launches a wave packs six rows and is gone; 5G NR BG1 Zc=384 HLTanhf32 +0.7 % over three alternating pairs, round 5)
  const Tiling pack_t = make_tiling(G, tile, 64, n, 256, std::min<uint32_t>(target_waves, kPackWaves));
  auto pack = [&](const T *soft) {
    dev::pack_hard_kernel<T><<<pack_t.blocks, pack_t.threads, 0, s>>>(soft, w.hardbits, w.n_active, w.n_slots, n, tile,
                                                                      W, pack_t.sched.waves_per_chunk);
  };

  auto emit = [&](int zero_fill, int retire_only) {
    dim3 grid(std::min<uint32_t>((n + 63) / 64, retire_only ? kRetireBlocks : 4096), W);
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
  const Tiling mv_t = make_tiling(G, tile, 64, n, 256, kMoveWaves);
  uint32_t post_move_rows = 0;  // 0 = all rows; set by the flooding L-free paths below
  auto compact = [&](uint32_t remaining, T *msg_cur, bool with_chan, uint32_t msg_rows) {
    grp::compact_plan(s, ticked(max_iterations - remaining), w.plan, w.perm, w.slot_tmp, w.fill_cw, remaining,
        dev::CompactRule{kCompactHorizon, kCompactCostLive, kCompactCostSlots, kCompactMinFreedQ});
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
    grp::compact_commit(s, st, w.plan, w.unsat0, w.unsat1, w.n_slots, w.fill_cw, G);
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
    const uint32_t stream_block = kStreamBlock;
    const Tiling vn_t = make_tiling(G, tile, 64 * vec, n, stream_block, opt_waves_ ? opt_waves_ : kVnWaves);
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
    const uint32_t cn_reg = (streaming || !opt_cn_reg_ || d_row_recs_ == nullptr || impl_.rule != Rule::Tanh || impl_.fast ||
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
      const uint32_t wv = opt_waves_ ? opt_waves_ : kVnWaves;
      vn_keep_t = make_tiling(G, tile, 64 * vec, n_keep_, stream_block, wv);
      vn_keep_t.sched.reverse = 1u;  // tiles last to first: the ones the check-node pass wrote last are still in the Infinity Cache (+0.3 %)
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
          Launch<T>::template cn_minsum<true>(vec, wide_mask, cn_t, s, g, stp, chan, msg, unsat_out);
        else
          Launch<T>::template cn_minsum<false>(vec, wide_mask, cn_t, s, g, stp, post, msg, unsat_out);
      } else {
        if (first)
          Launch<T>::template cn_staged<true>(impl_.rule, cn_reg, d_row_recs_, cn_t, st_lds, s, g, stp, chan, msg, unsat_out,
                                              max_row_weight_);
        else
          Launch<T>::template cn_staged<false>(impl_.rule, cn_reg, d_row_recs_, cn_t, st_lds, s, g, stp, post, msg, unsat_out,
                                               max_row_weight_);
      }
      timed_end(kKernelCheck, s);
      timed_begin(kKernelVar, s);
      // (deferred L-free stores: the first convergences of a slice get their L-free posteriors from the records of the latched
      // iteration INSIDE this launch -- rounds 3-4 ran a small vn_free_rec_kernel launch behind it in every iteration, which
      // almost always found nothing: 4.4 us + a 5.7 us dispatch gap per iteration)
      if (quiet && it > 1 && opt_vn_event_) {
        const dev::VnEvent<T> ev{d_free_var_, d_free_rs_, rbuf[(it - 1) & 1], n_free_};
        Launch<T>::vn_event(vec, rec_w_, vn_keep_t, s, g_keep, st, chan, m_out, post, unsat_out, unsat[(it + 1) & 1],
                            static_cast<int32_t>(it) - 1, ev);
      } else {
        Launch<T>::vn(lfree, vec, lfree ? vn_keep_t : vn_t, s, lfree ? g_keep : g, st, chan, m_out, post,
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
      Launch<T>::vn(true, vec, vn_free_t, s, g_free, st_nolatch, chan, mbuf[max_iterations & 1], post,
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
    // (min-sum streams its rows whatever their length -- unless "staged_minsum" sends it through the LDS-staged kernel)
    const bool streaming = impl_.rule == Rule::Minsum && !opt_staged_minsum_;
    if (!staged_block(lds_columns, max_row_weight_, sizeof(T), &threads, &lds) && !streaming) {
      // some level has rows beyond the LDS: those levels keep their columns in HBM (a small launch, one region per wave)
      const size_t waves_bound = size_t(kScratchWaves) + size_t(G / 64) * (kScratchThreads / 64);
      if (int rc = ensure_row_scratch(w, waves_bound * 2 * max_row_weight_ * 64 * sizeof(T))) return rc;
    }
    const uint32_t n_levels = level_ptr_.empty() ? 0 : static_cast<uint32_t>(level_ptr_.size() - 1);
    const dev::State st0 = st;
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
            Launch<T>::template hl_minsum<true>(vec, t, s, g, st, d_level_rows_ + r0, cnt, post, msg);
          else
            Launch<T>::template hl_minsum<false>(vec, t, s, g, st, d_level_rows_ + r0, cnt, post, msg);
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

}  // namespace ldpc
