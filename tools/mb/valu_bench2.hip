// Second issue-rate micro-benchmark (tools only): what do selects, compares, SGPR / literal operands
// and conversions cost next to a plain v_fma_f32 on gfx950?  valu_bench.hip measured v_cndmask_b32 at
// 1/8 of the v_fma_f32 rate; this one separates the possible causes (VCC vs SGPR-pair mask, SGPR and
// literal operands of ordinary instructions, compare+select pairs as the compiler emits them) and
// checks whether a slow instruction overlaps with fast ones issued between.
//   hipcc -O3 --offload-arch=gfx950 tools/mb/valu_bench2.hip -o tools/mb/valu_bench2
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
  X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int reps, float sb, float sc) {
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; i++) a[i] = 1.0f + 1e-3f * float(threadIdx.x + i);
  float b = 1.0001f + out[0] * 0.f, c = 1e-7f;
  unsigned long long m = 0x5555555555555555ull;  // a mask in an SGPR pair
  asm volatile("s_mov_b64 %0, %0" : "+s"(m));
  for (int r = 0; r < reps; r++) {
#define ONE(i)                                                                                                   \
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                         \
  if (KIND == 1) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                        \
  if (KIND == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m));                 \
  if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));                        \
  if (KIND == 4) asm volatile("v_mul_f32_e32 %0, 0x3f800347, %0" : "+v"(a[i]));                                  \
  if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "s"(sb));                       \
  if (KIND == 6) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); \
  if (KIND == 7) asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21"); \
  if (KIND == 8) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
  if (KIND == 9) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                        \
  if (KIND == 10) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));                        \
  if (KIND == 11) asm volatile("v_cvt_f32_i32_e32 %0, %0" : "+v"(a[i]));                                         \
  if (KIND == 12) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]));                                      \
  if (KIND == 13) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
  if (KIND == 14) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
  if (KIND == 15) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(b));                                    \
  if (KIND == 16) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0" : : "v"(a[i]), "v"(b) : "vcc");                    \
  if (KIND == 17) asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %0" : : "v"(a[i]), "v"(b) : "s20", "s21");        \
  if (KIND == 18) { if ((i & 3) == 0) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } \
  if (KIND == 19) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));                                             \
  if (KIND == 20) { if ((i & 3) == 0) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } \
  if (KIND == 21) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(a[i]));                                         \
  if (KIND == 22) asm volatile("v_and_b32_e32 %0, 0x7fffff, %0" : "+v"(a[i]));                                   \
  if (KIND == 23) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); \
  if (KIND == 24) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    REP32(ONE)
#undef ONE
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + float(m);
}

int main() {
  const int blocks = 256 * 8, threads = 256, reps = 1000;
  float *out;
  (void)hipMalloc(&out, size_t(blocks) * threads * 4);
  (void)hipMemset(out, 0, size_t(blocks) * threads * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double base = 0;
  auto time = [&](const char *name, auto kern, double insts_per_slot) {
    float best = 1e9f;
    for (int w = 0; w < 3; w++) {
      (void)hipEventRecord(e0);
      kern<<<blocks, threads>>>(out, reps, 1.0001f, 1e-7f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double slots = double(blocks) * (threads / 64) * reps * 32.0;
    const double ns_per_slot_cu = best * 1e6 / (slots / 256.0);
    if (base == 0) base = best;
    printf("%-44s %8.3f ms  %6.3f ns per slot per CU  = %5.2f x v_fma_f32  (%g instruction(s) per slot)\n", name, best,
           ns_per_slot_cu, best / base, insts_per_slot);
  };
  time("v_fma_f32 v,v,v,v", k<0>, 1);
  time("v_cndmask_b32_e32 (vcc)", k<1>, 1);
  time("v_cndmask_b32_e64 (sgpr pair)", k<2>, 1);
  time("v_fma_f32 v,v,s,v (one SGPR operand)", k<3>, 1);
  time("v_mul_f32 v, literal, v", k<4>, 1);
  time("v_fma_f32 v,v,s,s (same SGPR twice)", k<5>, 1);
  time("v_cmp_lt_f32 vcc + v_cndmask vcc", k<6>, 2);
  time("v_cmp_lt_f32 s[..] + v_cndmask s[..]", k<7>, 2);
  time("v_max_f32", k<8>, 1);
  time("v_med3_f32", k<9>, 1);
  time("v_bfi_b32", k<10>, 1);
  time("v_cvt_f32_i32", k<11>, 1);
  time("v_lshlrev_b32", k<12>, 1);
  time("v_add_u32", k<13>, 1);
  time("v_sub_f32", k<14>, 1);
  time("v_mov_b32", k<15>, 1);
  time("v_cmp_lt_f32_e32 vcc", k<16>, 1);
  time("v_cmp_lt_f32_e64 s[..]", k<17>, 1);
  time("1 v_cndmask(vcc) : 3 v_fma", k<18>, 1);
  time("v_rcp_f32", k<19>, 1);
  time("1 v_rcp : 3 v_fma", k<20>, 1);
  time("v_cvt_i32_f32", k<21>, 1);
  time("v_and_b32 literal", k<22>, 1);
  time("cmp + cndmask + 2 fma (dependent)", k<23>, 4);
  time("4 fma (dependent)", k<24>, 4);
  return 0;
}
