// The small-batch paths: one persistent launch per call (latency.hip.h, latency_edge.hip.h) and their kernels.
#include "device_decoder_internal.h"

namespace ldpc {

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
    lp.grid = static_cast<uint32_t>(std::min<int>(resident, 256));
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

}  // namespace ldpc
