// Standalone timing of the slice-persistent layered kernel (kernels.hip.h, hl_slice_kernel) on one code:
// random Qv / R of `batch` codewords in the decoder's tiled layout, `reps` launches of one iteration each.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/mb/slice_bench.hip -o tools/mb/slice_bench
//   tools/mb/slice_bench tools/mb/graph_bg1_384.bin [batch=8192] [slice=32] [threads=1024] [reps=10]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../ldpc_toolbox_amd/csrc/kernels.hip.h"
#include "../../ldpc_toolbox_amd/csrc/slice_tasks.h"

using namespace ldpc;

#ifndef RULE
#define RULE dev::kRuleTanh
#endif

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

template <typename V>
static uint32_t *up(const std::vector<V> &v) {
  uint32_t *d = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&d), std::max<size_t>(v.size(), 1) * sizeof(V)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const uint32_t batch = argc > 2 ? std::atoi(argv[2]) : 8192, slice = argc > 3 ? std::atoi(argv[3]) : 32,
                 threads = argc > 4 ? std::atoi(argv[4]) : 1024, reps = argc > 5 ? std::atoi(argv[5]) : 10,
                 tile = argc > 6 ? std::atoi(argv[6]) : 256;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  uint32_t hdr[3];
  if (std::fread(hdr, 4, 3, f) != 3) return 2;
  const uint32_t m = hdr[0], n = hdr[1], E = hdr[2];
  std::vector<uint32_t> row_ptr(m + 1), edge_col(E);
  if (std::fread(row_ptr.data(), 4, m + 1, f) != m + 1 || std::fread(edge_col.data(), 4, E, f) != E) return 2;
  std::fclose(f);
  uint32_t dmax = 0;
  for (uint32_t r = 0; r < m; r++) dmax = std::max(dmax, row_ptr[r + 1] - row_ptr[r]);
  dmax = std::max(dmax, 10u);  // (the rare-argument fix-up parks up to ten values in a lane's column)
  const LevelTables lv = build_levels(row_ptr, edge_col, m, n);
  const SliceTasks st = build_slice_tasks(lv, row_ptr, edge_col, 64 / slice, RULE == dev::kRuleTanh);
  const uint32_t n_levels = uint32_t(lv.maxdeg.size());
  if (!st.fits) std::printf("note: some rows fit no task of this slice width\n");
  edge_col.resize(E + 16, 0);
  const uint32_t G = batch;
  float *Q, *R;
  CK(hipMalloc(reinterpret_cast<void **>(&Q), size_t(n) * G * 4));
  CK(hipMalloc(reinterpret_cast<void **>(&R), size_t(E) * G * 4));
  {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> h(size_t(1) << 22);
    for (auto &x : h) x = 3.f * nd(rng);
    for (size_t off = 0; off < size_t(n) * G; off += h.size())
      CK(hipMemcpy(Q + off, h.data(), std::min(h.size(), size_t(n) * G - off) * 4, hipMemcpyHostToDevice));
    for (auto &x : h) x *= 0.3f;
    for (size_t off = 0; off < size_t(E) * G; off += h.size())
      CK(hipMemcpy(R + off, h.data(), std::min(h.size(), size_t(E) * G - off) * 4, hipMemcpyHostToDevice));
  }
  uint32_t *flags;
  CK(hipMalloc(reinterpret_cast<void **>(&flags), (size_t(G) * 2 + 128) * 4));
  CK(hipMemset(flags, 0, (size_t(G) * 2 + 128) * 4));
  uint32_t *done = flags, *n_active = flags + 2 * G, *n_slots = flags + 2 * G + 1;
  const uint32_t init[2] = {G, G};
  CK(hipMemcpy(n_active, init, 8, hipMemcpyHostToDevice));
  dev::Graph g{up(row_ptr), up(edge_col), nullptr, nullptr, m, n, E, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  dev::State s{done, reinterpret_cast<int32_t *>(flags + G), n_active, n_slots, nullptr, nullptr, 0, 0, nullptr, nullptr, 0};
  uint32_t *d_tasks = up(st.tasks), *d_tptr = up(st.task_ptr);
  const uint32_t columns = (RULE == dev::kRuleTanh || RULE == dev::kRuleTanhFast) ? 1 : 2;
  const size_t lds = (size_t(columns) * dmax * 4 + 2 * 10 * 4) * threads + 16;
  auto launch = [&](auto k) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    k<<<G / slice, threads, lds, 0>>>(g, s, d_tasks, d_tptr, n_levels, tile, Q, R, dmax, columns);
  };
  auto run = [&]() {
    if (slice == 32 && threads == 1024)
      launch(dev::hl_slice_kernel<RULE, float, 32, 1024, false>);
    else if (slice == 32 && threads == 512)
      launch(dev::hl_slice_kernel<RULE, float, 32, 512, false>);
    else if (slice == 32 && threads == 768)
      launch(dev::hl_slice_kernel<RULE, float, 32, 768, false>);
    else
      launch(dev::hl_slice_kernel<RULE, float, 64, 1024, false>);
  };
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  run();
  run();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (uint32_t i = 0; i < reps; i++) run();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  const double per = ms / reps, bytes = double(4.0 * E + n) * 4 * G;
  std::printf("tile %u batch %u slice %u threads %u levels %u tasks %u lds %zu: %.3f ms per iteration  -> %.0f cw/s at 50 it  frac %.3f\n", tile, batch,
              slice, threads, n_levels, st.task_ptr.back(), lds, per, G / (per * 50e-3), bytes / (per * 1e-3) / 8e12);
#ifdef SLICE_COUNT
  uint64_t clk[2];
  CK(hipMemcpy(clk, n_slots + 40, sizeof(clk), hipMemcpyDeviceToHost));
  std::printf("block 0: %llu shader cycles in %llu ticks of the 100 MHz counter -> %.0f MHz\n", (unsigned long long)clk[0],
              (unsigned long long)clk[1], clk[0] / (clk[1] / 100.0));
  uint32_t cnt[32];
  CK(hipMemcpy(cnt, n_slots + 2, sizeof(cnt), hipMemcpyDeviceToHost));
  for (int i = 0; i < 32; i++) std::printf("level %d: %u tasks done (%u per launch; table %u)\n", i, cnt[i], cnt[i] / (reps + 2) / (G / slice), st.task_ptr[i + 1] - st.task_ptr[i]);
#endif
  return 0;
}
