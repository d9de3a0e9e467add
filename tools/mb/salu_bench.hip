// Does scalar-instruction load slow the vector issue of a CU?  (tools only; not the product)
//   hipcc -O3 --offload-arch=gfx950 tools/mb/salu_bench.hip -o tools/mb/salu_bench
// Every wave runs REPS x (32 dependent-free v_fma_f32 interleaved with M scalar instructions); 8 waves per SIMD on every CU.
// Reported: time, and vector instructions per cycle per SIMD, for M = 0, 8, 16, 32, 48, 64 (s_add_u32 on private SGPRs) and
// for M branch-region pairs (s_and_saveexec_b64 + s_or_b64, exec unchanged).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int M, int KIND>
__global__ __launch_bounds__(256, 8) void k(float *out, int reps) {
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; i++) a[i] = 1.0f + 1e-3f * float(threadIdx.x + i);
  const float b = 1.0001f, c = 1e-7f;
  unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      // M scalar instructions spread over the 32 vector ones
#pragma unroll
      for (int j = 0; j < (M * (i + 1)) / 32 - (M * i) / 32; j++) {
        if (KIND == 0) {
          asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
          asm volatile("" : "+s"(s1), "+s"(s2), "+s"(s3));
        } else {
          asm volatile("s_and_saveexec_b64 s[10:11], exec\n\ts_or_b64 exec, exec, s[10:11]" : : : "s10", "s11", "scc");
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + float(s0);
}

template <int M, int KIND>
void run(float *out, const char *what) {
  const int reps = 2000, blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<M, KIND><<<blocks, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<M, KIND><<<blocks, 256>>>(out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double vinst = double(blocks) * 4 * reps * 32;  // wave instructions
  printf("%-28s M = %2d per 32 v_fma: %8.3f ms   %.3f v_fma per ns chip-wide   (%.2f cycles per v_fma per SIMD at 2.4 GHz)\n", what, KIND == 0 ? M : 2 * M, ms,
         vinst / (ms * 1e6), ms * 1e-3 * 2.4e9 / (vinst / 1024));
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<0, 0>(out, "s_add_u32");
  run<8, 0>(out, "s_add_u32");
  run<16, 0>(out, "s_add_u32");
  run<32, 0>(out, "s_add_u32");
  run<48, 0>(out, "s_add_u32");
  run<64, 0>(out, "s_add_u32");
  run<4, 1>(out, "saveexec + or (pairs)");
  run<8, 1>(out, "saveexec + or (pairs)");
  run<16, 1>(out, "saveexec + or (pairs)");
  return 0;
}
