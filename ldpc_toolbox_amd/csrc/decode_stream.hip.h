// Continuous batching (DeviceDecoder::decode_stream): -DLDPC_EXPERIMENTS builds only; compiled with the f32 kernels
// (run_group_f32.hip).  DESIGN.md section 4.4: exact, and slower than drained batches in this layout.
#pragma once
#include "launch.hip.h"

namespace ldpc {

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
  const Tiling vn_keep_t = make_tiling(Gu, tile, 64 * vec, n_keep_, stream_block, kVnWaves);

  // every slot starts empty: finished, no codeword
  grp::init_group(s, w.done, w.iters, w.unsat0, w.unsat1, w.n_active, w.n_slots, w.slot_cw,
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
  const uint32_t every = kStreamHarvest;
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
    Launch<T>::vn(true, vec, vn_keep_t, s, g_keep, st, chan, msg, post, unsat_out, unsat[(it + 1) & 1],
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

}  // namespace ldpc
