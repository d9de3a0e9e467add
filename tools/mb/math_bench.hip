// Micro-benchmark of the glibc-identical device functions on decoder-like argument distributions:
// ns per call per lane-instance, to see which ones carry the transcendental rules' time.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/mb/math_bench.hip -o /tmp/math_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "../../ldpc_toolbox_amd/csrc/exact_math.h"
using namespace ldpc;

template <int F>
__global__ void k(const float *in, float *out, int n, int reps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = in[i], acc = 0.f;
  for (int r = 0; r < reps; r++) {
    float y;
    if (F == 0) y = em::tanhf(x);
    else if (F == 1) y = em::atanh_rs(x);   // atanh as Rust computes it
    else if (F == 2) y = em::log1pf(x);
    else if (F == 3) y = em::expf(x);
    else if (F == 4) y = em::logf(x);
    else if (F == 5) y = em::log1pf(em::expf(-fabsf(x)));
    else if (F == 6) y = em::expm1f(x);
    else y = x * 1.0001f + 0.5f;
    acc += y;
    x = x + 1e-7f * y;   // keep the loop from being hoisted; stays in the same range
  }
  out[i] = acc;
}

int main() {
  const int n = 1 << 22, reps = 64;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> h(n);
  float *d_in, *d_out;
  hipMalloc(&d_in, n * 4); hipMalloc(&d_out, n * 4);
  const char *names[] = {"tanhf(h), h~N(0,3) clamped +-9", "atanh_rs(p), p=prod of tanh", "log1pf(y), y in (0,1]", "expf(-a), a~|N(0,4)|",
                         "logf(t), t in (0,1]", "log1pf(expf(-|a|))", "expm1f(x), x in (-2,18)", "baseline fma"};
  for (int f = 0; f < 8; f++) {
    for (int i = 0; i < n; i++) {
      float v;
      switch (f) {
        case 0: v = std::max(-9.f, std::min(9.f, 3.f * nd(rng))); break;
        case 1: { float p = 1.f; for (int j = 0; j < 5; j++) p *= std::tanh(std::max(-9.f, std::min(9.f, 3.f * nd(rng)))); v = p; break; }
        case 2: v = ud(rng); break;
        case 3: v = -std::fabs(4.f * nd(rng)); break;
        case 4: v = std::max(1e-6f, ud(rng)); break;
        case 5: v = 4.f * nd(rng); break;
        case 6: v = (ud(rng) < 0.5f) ? -2.f * ud(rng) : 2.f + 16.f * ud(rng); break;
        default: v = nd(rng);
      }
      h[i] = v;
    }
    hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&]() {
      switch (f) {
        case 0: k<0><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 1: k<1><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 2: k<2><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 3: k<3><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 4: k<4><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 5: k<5><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 6: k<6><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        default: k<7><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
      }
    };
    run(); hipDeviceSynchronize();
    hipEventRecord(a); run(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double calls = double(n) * reps;
    // SIMD-cycles per wave-call: ms * 1024 SIMDs * 2.4e6 cycles/ms / (calls / 64)
    printf("%-36s %8.3f ms  %7.2f ps/call  ~%6.0f SIMD-cycles per wave call\n", names[f], ms, ms * 1e9 / calls,
           ms * 1024 * 2.4e6 / (calls / 64));
  }
  return 0;
}
