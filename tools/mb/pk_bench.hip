// throughput of the scalar and the two-per-lane forms of the Tanh rule's functions (exact_math.h) on the device
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/mb/pk_bench.hip -o tools/mb/pk_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "exact_math_x2.h"
using namespace ldpc;

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
  extern __shared__ float occupancy_limiter[];  // dynamic LDS only caps the workgroups per CU
  if (iters < 0) occupancy_limiter[threadIdx.x] = seed;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float a = seed + 1e-4f * (i & 1023), b = -seed * 0.5f + 2e-4f * (i & 511);
  float acc0 = 0.f, acc1 = 0.f;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // scalar, two values
      const float t0 = em::tanhf_c9(a), t1 = em::tanhf_c9(b);
      const float u0 = em::atanh_rs(t0 * 0.75f), u1 = em::atanh_rs(t1 * 0.5f);
      acc0 += u0; acc1 += u1;
    } else if (MODE == 1) {  // pair
      const em::f32x2 t = em::tanhf_c9(em::f32x2{a, b});
      const em::f32x2 u = em::atanh_rs(t * em::f32x2{0.75f, 0.5f});
      acc0 += u.x; acc1 += u.y;
    } else if (MODE == 2) {  // pure packed fma chain
      em::f32x2 v{a, b};
      for (int j = 0; j < 64; j++) v = em::fma2(v, em::f32x2{0.999f, 1.001f}, em::f32x2{1e-3f, -1e-3f});
      acc0 += v.x; acc1 += v.y;
    } else if (MODE == 4) {  // scalar, one value per thread
      const float t0 = em::tanhf_c9(a);
      const float u0 = em::atanh_rs(t0 * 0.75f);
      acc0 += u0;
    } else if (MODE == 5) {  // scalar fma chain, one value
      float v0 = a;
      for (int j = 0; j < 64; j++) v0 = __builtin_fmaf(v0, 0.999f, 1e-3f);
      acc0 += v0;
    } else {  // pure scalar fma chain x2
      float v0 = a, v1 = b;
      for (int j = 0; j < 64; j++) { v0 = __builtin_fmaf(v0, 0.999f, 1e-3f); v1 = __builtin_fmaf(v1, 1.001f, -1e-3f); }
      acc0 += v0; acc1 += v1;
    }
    a += 1e-3f; b -= 1e-3f;
    if (a > 8.f) a -= 16.f;
    if (b < -8.f) b += 16.f;
  }
  out[i] = acc0 + acc1;
}

// --check: the two-per-lane forms against the scalar ones (themselves checked against glibc on every float by
// tools/check_exact_math_device.hip), all 2^32 arguments, both positions, the partner element from another class
__global__ void check_kernel(uint32_t base, unsigned long long *bad) {
  const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
  const float x = em::as_f32(bits);
  const float p = em::as_f32((bits * 2654435761u) ^ (bits >> 7));
  unsigned long long n = 0;
  auto same = [](float a, float b) { return em::as_u32(a) == em::as_u32(b) || (a != a && b != b); };
  const em::f32x2 l0 = em::log1pf(em::f32x2{x, p}), l1 = em::log1pf(em::f32x2{p, x});
  n += !same(l0.x, em::log1pf(x)) + !same(l1.y, em::log1pf(x));
  const em::f32x2 a0 = em::atanh_rs(em::f32x2{x, p}), a1 = em::atanh_rs(em::f32x2{p, x});
  n += !same(a0.x, em::atanh_rs(x)) + !same(a1.y, em::atanh_rs(x));
  if (fabsf(x) <= 9.0f) {
    const float q = fabsf(p) <= 9.0f ? p : em::as_f32((em::as_u32(p) & 0x807fffffu) | 0x40000000u);
    const em::f32x2 t0 = em::tanhf_c9(em::f32x2{x, q}), t1 = em::tanhf_c9(em::f32x2{q, x});
    n += !same(t0.x, em::tanhf_c9(x)) + !same(t1.y, em::tanhf_c9(x));
  }
  if (n) atomicAdd(bad, n);
}

template <int MODE>
void run(const char *name, float *d, int iters, int per_thread = 2, int waves_per_simd = 8) {
  const size_t lds = waves_per_simd >= 8 ? 0 : (size_t(160) * 1024 / waves_per_simd) - 512;
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8 * 4;
  k<MODE><<<blocks, 256, lds>>>(d, 4, 1.0f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256, lds>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double pairs = double(blocks) * 256 * iters;
  printf("%-28s %d waves/SIMD %8.3f ms  %8.2f G element-evaluations/s\n", name, waves_per_simd, ms, per_thread * pairs / ms / 1e6);
}
int main(int argc, char **argv) {
  if (argc > 1) {
    unsigned long long *bad, h = 0;
    hipMalloc(&bad, 8);
    hipMemset(bad, 0, 8);
    for (uint64_t base = 0; base < (1ull << 32); base += 1u << 28) check_kernel<<<(1u << 28) / 256, 256>>>(uint32_t(base), bad);
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("two-per-lane tanhf_c9 / log1pf / atanh against the scalar forms, 2^32 arguments, both positions: %llu mismatches\n", h);
    return h != 0;
  }
  float *d; hipMalloc(&d, 256 * 8 * 4 * 256 * 4);
  for (int w : {1, 2, 3, 4, 6}) {
    run<4>("tanh+atanh scalar x1", d, 500, 1, w);
    run<0>("tanh+atanh scalar x2", d, 500, 2, w);
    run<1>("tanh+atanh pair", d, 500, 2, w);
  }
  run<0>("tanh+atanh scalar x2", d, 2000);
  run<1>("tanh+atanh pair", d, 2000);
  run<4>("tanh+atanh scalar x1", d, 2000, 1);
  run<5>("fma chain scalar x1 (64)", d, 500, 1);
  run<3>("fma chain scalar x2 (64)", d, 500);
  run<2>("fma chain pair (64)", d, 500);
  return 0;
}
