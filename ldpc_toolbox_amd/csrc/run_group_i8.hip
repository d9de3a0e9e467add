// The 8-bit quantised rules: DeviceDecoder::run_group_i8 and every kernel it launches (kernels_i8.hip.h).
#define LDPC_I8_KERNELS_TU 1  // this translation unit compiles the 8-bit rules' one non-template kernel
#include "device_decoder_internal.h"
#include "kernels_i8.hip.h"

namespace ldpc {

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

  grp::init_group(s, w.done, w.iters, w.unsat0, w.unsat1, w.n_active, w.n_slots,
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
      std::max<uint32_t>(1, std::min<uint32_t>(64, uint32_t(uint64_t(m) * synd_chunks * 64 / kSyndThreads)));
  const uint32_t synd_threads = 64 * synd_chunks * ((m + synd_rows - 1) / synd_rows);
  auto syndrome_of = [&](const uint64_t *hard, uint32_t *unsat) {
    if (m == 0) return;
    grp::syndrome_bits(s, synd_threads, d_row_ptr_, d_edge_col_, m, hard, unsat,
                                                                         w.n_active, w.n_slots, W, synd_rows);
  };
  auto latch = [&](uint32_t *unsat, int32_t it) {
    grp::latch(s, w.done, w.iters, unsat, w.n_active, it, G);
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

}  // namespace ldpc
