// The float rules' kernel launches: every template choice (pack width, mask width, loads in flight, record words, rule,
// degree bucket) is made here, from run-time values.  Included by run_group.hip.h only: a translation unit that includes this
// header and calls into Launch<T> instantiates -- compiles -- T's kernels.
#pragma once
#include "device_decoder_internal.h"

namespace ldpc {

template <typename T>
struct Launch {
  // flooding min-sum check nodes: VEC x mask width x unroll x FIRST
  template <int VEC, typename MASK, bool FIRST>
  static void cn_minsum_u(const Tiling &t, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const T *L, T *msg, uint32_t *unsat) {
    // (eight loads in flight, nontemporal messages: the four-load and the cached-message variants were tuning knobs within
    // a percent of these, gone in round 6)
    dev::cn_minsum_kernel<T, VEC, MASK, 8, FIRST, true><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, L, msg, unsat);
  }
  template <int VEC, bool FIRST>
  static void cn_minsum_m(bool wide_mask, const Tiling &t, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const T *L, T *msg, uint32_t *unsat) {
    if (wide_mask)
      cn_minsum_u<VEC, uint64_t, FIRST>(t, s, g, st, L, msg, unsat);
    else
      cn_minsum_u<VEC, uint32_t, FIRST>(t, s, g, st, L, msg, unsat);
  }
  // L-free variant (double-buffered messages)
  template <int VEC, typename MASK, bool FIRST>
  static void cn_lfree_u(const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st, const T *chan,
                         T *post, const T *msg_in, T *msg_out, uint32_t *unsat) {
    // (four loads in flight, nontemporal stores, cached loads of the previous messages: what round 2 settled on)
    dev::cn_minsum_lfree_kernel<T, VEC, MASK, 4, FIRST, true, false><<<t.blocks, t.threads, 0, s>>>(
        g, t.sched, st, chan, post, msg_in, msg_out, unsat);
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
  static void cn_minsum(uint32_t vec, bool wide_mask, const Tiling &t, hipStream_t s,
                        const dev::Graph &g, const dev::State &st, const T *L, T *msg, uint32_t *unsat) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      cn_minsum_m<kMaxVec, FIRST>(wide_mask, t, s, g, st, L, msg, unsat);
    else if (vec >= 2)
      cn_minsum_m<2, FIRST>(wide_mask, t, s, g, st, L, msg, unsat);
    else
      cn_minsum_m<1, FIRST>(wide_mask, t, s, g, st, L, msg, unsat);
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
    if constexpr (RULE == dev::kRuleTanh) {  // (the opt-in "@fast" variant keeps the LDS-staged kernel)
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
  static void vn_l(const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                   const T *chan, const T *msg, T *post, const uint32_t *unsat_in, uint32_t *unsat_clear,
                   int32_t latch_it) {
    // (eight loads in flight; the messages are read once: nontemporal -- 732 -> 680 us on DVB-S2 1/2 in round 2)
    dev::vn_kernel<T, VEC, 8, true, LIST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post, unsat_in, unsat_clear,
                                                                        latch_it);
  }
  template <int VEC>
  static void vn_v(bool list, const Tiling &t, hipStream_t s, const dev::Graph &g,
                   const dev::State &st, const T *chan, const T *msg, T *post, const uint32_t *unsat_in,
                   uint32_t *unsat_clear, int32_t latch_it) {
    if (list)
      vn_l<VEC, true>(t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
    else
      vn_l<VEC, false>(t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
  }
  static void vn(bool list, uint32_t vec, const Tiling &t, hipStream_t s, const dev::Graph &g,
                 const dev::State &st, const T *chan, const T *msg, T *post, const uint32_t *unsat_in,
                 uint32_t *unsat_clear, int32_t latch_it) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      vn_v<kMaxVec>(list, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
    else if (vec >= 2)
      vn_v<2>(list, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
    else
      vn_v<1>(list, t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it);
  }

  // the list variant that also rebuilds the L-free posteriors of a slice's first convergences (kernels_flooding.hip.h, EVW)
  template <int VEC, int EVW>
  static void vn_event_v(const Tiling &t, hipStream_t s, const dev::Graph &g, const dev::State &st,
                         const T *chan, const T *msg, T *post, const uint32_t *unsat_in, uint32_t *unsat_clear,
                         int32_t latch_it, const dev::VnEvent<T> &ev) {
    dev::vn_kernel<T, VEC, 8, true, true, EVW><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
  }
  static void vn_event(uint32_t vec, uint32_t recw, const Tiling &t, hipStream_t s, const dev::Graph &g,
                       const dev::State &st, const T *chan, const T *msg, T *post, const uint32_t *unsat_in,
                       uint32_t *unsat_clear, int32_t latch_it, const dev::VnEvent<T> &ev) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    auto go = [&](auto vecc) {
      constexpr int V = decltype(vecc)::value;
      if (recw == 3)
        vn_event_v<V, 3>(t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
      else
        vn_event_v<V, 4>(t, s, g, st, chan, msg, post, unsat_in, unsat_clear, latch_it, ev);
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
  static void hl_minsum_v(const Tiling &t, hipStream_t s, const dev::Graph &g,
                          const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R) {
    dev::hl_minsum_kernel<T, VEC, 8, FIRST><<<t.blocks, t.threads, 0, s>>>(g, t.sched, st, level_rows, n_level, Q, R);
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
  static void hl_minsum(uint32_t vec, const Tiling &t, hipStream_t s, const dev::Graph &g,
                        const dev::State &st, const uint32_t *level_rows, uint32_t n_level, T *Q, T *R) {
    constexpr int kMaxVec = sizeof(T) == 4 ? 4 : 2;
    if (vec == 4 && kMaxVec == 4)
      hl_minsum_v<kMaxVec, FIRST>(t, s, g, st, level_rows, n_level, Q, R);
    else if (vec >= 2)
      hl_minsum_v<2, FIRST>(t, s, g, st, level_rows, n_level, Q, R);
    else
      hl_minsum_v<1, FIRST>(t, s, g, st, level_rows, n_level, Q, R);
  }
};

}  // namespace ldpc
