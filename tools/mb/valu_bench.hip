// Issue-rate micro-benchmark of the vector instructions the transcendental rules are made of, to decide
// whether two codewords per lane through packed f32 instructions would pay (tools only; not the product).
//   hipcc -O3 --offload-arch=gfx950 tools/mb/valu_bench.hip -o tools/mb/valu_bench
// Every wave runs REPS x 32 independent instructions of one kind on 32 (or 16 register pairs of) VGPRs;
// reported: wave-instructions per cycle per SIMD at the clock the run held (s_memtime delta of wave 0).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int reps, unsigned long long *cycles) {
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; i++) a[i] = 1.0f + 1e-3f * float(threadIdx.x + i);
  const float b = 1.0001f, c = 1e-7f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
      if (KIND == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      if (KIND == 5) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND>
__global__ __launch_bounds__(256) void kp(float *out, int reps, unsigned long long *cycles) {
  float2v a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = float2v{1.0f + 1e-3f * float(threadIdx.x + i), 1.0f + 2e-3f * float(threadIdx.x + i)};
  const float2v b = {1.0001f, 1.0002f}, c = {1e-7f, 2e-7f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  const int blocks = 256 * 8, threads = 256, reps = 2000;  // 8 workgroups per CU: 8 waves per SIMD
  float *out;
  unsigned long long *cyc, h = 0;
  hipMalloc(&out, size_t(blocks) * threads * 4);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char *name, auto kern, double lane_ops_per_inst) {
    for (int w = 0; w < 2; w++) {
      hipEventRecord(e0);
      kern<<<blocks, threads>>>(out, reps, cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double insts = double(blocks) * (threads / 64) * reps * 32.0;  // wave-instructions
    const double per_s = insts / (ms * 1e-3);
    printf("%-16s %8.3f ms  %7.2f G wave-inst/s  = %6.3f wave-inst/ns/CU  %8.1f T lane-ops/s  (wave0: %llu ticks of s_memtime/100MHz?)\n",
           name, ms, per_s / 1e9, per_s / 1e9 / 256, per_s * 64 * lane_ops_per_inst / 1e12, h);
  };
  time("v_fma_f32", k<0>, 1);
  time("v_mul_f32", k<1>, 1);
  time("v_rcp_f32", k<2>, 1);
  time("v_cndmask_b32", k<3>, 1);
  time("v_cmp_lt_f32", k<4>, 1);
  time("v_cvt_i32_f32", k<5>, 1);
  time("v_and_b32", k<6>, 1);
  time("v_pk_fma_f32", kp<0>, 2);
  time("v_pk_mul_f32", kp<1>, 2);
  time("v_pk_add_f32", kp<2>, 2);
  return 0;
}
