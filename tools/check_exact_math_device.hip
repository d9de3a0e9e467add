// Device-side exhaustive check of csrc/exact_math.h: every one of the 2^32 float arguments is
// evaluated ON THE GPU (gfx950) and compared, bit for bit, with the host's glibc (all NaNs count as
// equal).  tools/check_exact_math.cpp does the same for the host build of the header; this one pins
// the device code generation (no FMA contraction, correctly rounded division and sqrt, denormals).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -pthread tools/check_exact_math_device.hip -o tools/mb/check_device
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../ldpc_toolbox_amd/csrc/exact_math.h"
using namespace ldpc;

enum { kExp, kLog, kLog1p, kExpm1, kTanh, kAtanhRs, kCorr, kTanhC9, kCorrF, kPhiF, kAtanhMain, kCount };
static const char *kNames[kCount] = {"expf", "logf", "log1pf", "expm1f", "tanhf", "atanh (Rust: 0.5*ln_1p(2x/(1-x)))",
                                     "ln_1p(exp(-|x|))", "tanhf_c9 (|x| <= 9; tanhf elsewhere)",
                                     "corrf(|x|) = fused ln_1p(exp(-|x|))",
                                     "phif(x) = fused -ln(tanh(max(x, 1e-30) / 2))",
                                     "atanh main path (atanh_rs_main; atanh_rs where it reports a rare argument)"};

__host__ __device__ inline float eval_mine(int f, float x) {
  switch (f) {
    case kExp: return em::expf(x);
    case kLog: return em::logf(x);
    case kLog1p: return em::log1pf(x);
    case kExpm1: return em::expm1f(x);
    case kTanh: return em::tanhf(x);
    case kAtanhRs: return em::atanh_rs(x);
    case kTanhC9: return (fabsf(x) <= 9.0f) ? em::tanhf_c9(x) : em::tanhf(x);
    case kCorrF: return em::corrf(fabsf(x));
    case kPhiF: return em::phif(x);
    case kAtanhMain: {
      bool rare = false;
      const float y = em::atanh_rs_main(x, &rare);
      return rare ? em::atanh_rs(x) : y;
    }
    default: return em::log1pf(em::expf(-fabsf(x)));
  }
}
static float eval_ref(int f, float x) {
  switch (f) {
    case kExp: return ::expf(x);
    case kLog: return ::logf(x);
    case kLog1p: return ::log1pf(x);
    case kExpm1: return ::expm1f(x);
    case kTanh: return ::tanhf(x);
    case kAtanhRs: return 0.5f * ::log1pf((2.0f * x) / (1.0f - x));
    case kTanhC9: return ::tanhf(x);
    case kCorrF: return ::log1pf(::expf(-fabsf(x)));
    case kPhiF: return -(::logf(::tanhf(0.5f * ::fmaxf(x, 1e-30f))));
    case kAtanhMain: return 0.5f * ::log1pf((2.0f * x) / (1.0f - x));
    default: return ::log1pf(::expf(-fabsf(x)));
  }
}

__global__ void eval_kernel(int f, uint32_t base, uint32_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = em::as_u32(eval_mine(f, em::as_f32(base + i)));
}

int main(int argc, char **argv) {
  const uint32_t chunk = 1u << 26;
  uint32_t *d_out;
  if (hipMalloc(&d_out, size_t(chunk) * 4) != hipSuccess) return 2;
  std::vector<uint32_t> got(chunk);
  const unsigned nthreads = std::max(1u, std::thread::hardware_concurrency());
  int bad_functions = 0;
  for (int f = 0; f < kCount; f++) {
    if (argc > 1 && strstr(kNames[f], argv[1]) == nullptr) continue;
    std::atomic<unsigned long long> mism{0};
    std::atomic<uint32_t> first_bad{0};
    for (uint64_t base = 0; base < (1ull << 32); base += chunk) {
      eval_kernel<<<chunk / 256, 256>>>(f, static_cast<uint32_t>(base), d_out);
      if (hipMemcpy(got.data(), d_out, size_t(chunk) * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
          unsigned long long local = 0;
          for (uint64_t i = t; i < chunk; i += nthreads) {
            const uint32_t bits = static_cast<uint32_t>(base + i);
            const float want = eval_ref(f, em::as_f32(bits)), have = em::as_f32(got[i]);
            if (em::as_u32(want) != got[i] && !(want != want && have != have)) {
              if (local == 0 && mism.load() == 0) first_bad = bits;
              local++;
            }
          }
          mism += local;
        });
      for (auto &x : th) x.join();
    }
    printf("%-36s device vs glibc, 2^32 arguments: %llu mismatches", kNames[f], mism.load());
    if (mism.load()) {
      const float x = em::as_f32(first_bad.load());
      printf("   e.g. x=%a: glibc %a", x, eval_ref(f, x));
      bad_functions++;
    }
    printf("\n");
    fflush(stdout);
  }
  return bad_functions ? 1 : 0;
}
