// Micro-benchmark of the double-precision glibc-identical device functions (see math_bench.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "../../ldpc_toolbox_amd/csrc/exact_math.h"
using namespace ldpc;

template <int F>
__global__ void k(const double *in, double *out, int n, int reps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = in[i], acc = 0.0;
  for (int r = 0; r < reps; r++) {
    double y;
    if (F == 0) y = em::tanh(x);
    else if (F == 1) y = em::log(x);
    else if (F == 2) y = em::log1p(x);
    else if (F == 3) y = em::exp(x);
    else if (F == 4) y = em::expm1(x);
    else if (F == 5) y = -em::log(em::tanh(0.5 * (x > 1e-30 ? x : 1e-30)));   // phi
    else y = x * 1.0001 + 0.5;
    acc += y;
    x = x + 1e-13 * y;
  }
  out[i] = acc;
}

int main() {
  const int n = 1 << 21, reps = 32;
  std::mt19937 rng(1);
  std::normal_distribution<double> nd(0., 1.);
  std::uniform_real_distribution<double> ud(0., 1.);
  std::vector<double> h(n);
  double *d_in, *d_out;
  hipMalloc(&d_in, n * 8); hipMalloc(&d_out, n * 8);
  const char *names[] = {"tanh(h), h~N(0,3) clamped +-18", "log(t), t in (0,1]", "log1p(y), y in (0,1]", "exp(-a), a~|N(0,4)|",
                         "expm1(x), x in (-2,18)", "phi(x) = -log(tanh(x/2)), x~|N(0,6)|", "baseline fma"};
  for (int f = 0; f < 7; f++) {
    for (int i = 0; i < n; i++) {
      double v;
      switch (f) {
        case 0: v = std::max(-18., std::min(18., 3. * nd(rng))); break;
        case 1: v = std::max(1e-9, ud(rng)); break;
        case 2: v = ud(rng); break;
        case 3: v = -std::fabs(4. * nd(rng)); break;
        case 4: v = (ud(rng) < 0.5) ? -2. * ud(rng) : 2. + 16. * ud(rng); break;
        case 5: v = std::fabs(6. * nd(rng)); break;
        default: v = nd(rng);
      }
      h[i] = v;
    }
    hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&]() {
      switch (f) {
        case 0: k<0><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 1: k<1><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 2: k<2><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 3: k<3><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 4: k<4><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        case 5: k<5><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
        default: k<6><<<n / 256, 256>>>(d_in, d_out, n, reps); break;
      }
    };
    run(); hipDeviceSynchronize();
    hipEventRecord(a); run(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double calls = double(n) * reps;
    printf("%-40s %8.3f ms  %7.2f ps/call  ~%6.0f SIMD-cycles per wave call\n", names[f], ms, ms * 1e9 / calls,
           ms * 1024 * 2.4e6 / (calls / 64));
  }
  return 0;
}
