// The fold step of the 8-bit rules (Minstarapproxi8: acc = max(min(v, acc) - T[|v - acc|], 0), /root/reference/src/decoder/arithmetic.rs:741,
// T = round(8 ln(1 + e^(-t/8)))), three ways -- round 5's review asked for measurements instead of an instruction count:
//   form 0  today's (kernels_i8.hip.h, i8_minstar): per byte  bfe, sad_u8, min, min(t, 31), shift of a constant, popcount, saturating subtract
//   form 1  packed 16-bit: the word's bytes as two registers of two u16 lanes; min / |difference| with v_pk_*_u16, the table
//           as three 8-entry pieces looked up with v_perm_b32 and merged with v_pk_max_u16 (T is non-increasing), v_pk_sub_u16 clamp
//   form 2  a 16 KB table [acc][v] of the whole step in LDS: per byte  bfe, shift-or, ds_read_u8
// The loop is the check-node kernel's inner loop in miniature: a lane owns four codewords (one packed word), the row's d packed
// magnitudes sit in the lane's LDS column [slot][thread] and are folded again and again into four running values.
// Output: microseconds per 10^9 byte-steps on the whole chip and a checksum (equal across the forms).
// Build: hipcc -O3 --offload-arch=gfx950 i8_fold_bench.hip -o i8_fold_bench ; static instruction mix: tools/mb/isa_loops.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr uint32_t kBits = (1u << 0) | (1u << 2) | (1u << 4) | (1u << 8) | (1u << 12) | (1u << 21);
__host__ __device__ constexpr int table_entry(uint32_t t) { return int(t < 1) + int(t < 3) + int(t < 5) + int(t < 9) + int(t < 13) + int(t < 22); }
__device__ __forceinline__ uint32_t step_popcount(uint32_t v, uint32_t a) {
  const uint32_t t = __builtin_amdgcn_sad_u8(v, a, 0u);
  return __builtin_elementwise_sub_sat(min(v, a), static_cast<uint32_t>(__builtin_popcount(kBits >> min(t, 31u))));
}

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t as_u32(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ u16x2 as_pk(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
// two byte values per register (u16 lanes): one fold step for both
__device__ __forceinline__ uint32_t step_packed(uint32_t v2, uint32_t a2, uint32_t p0lo, uint32_t p0hi, uint32_t p1lo, uint32_t p1hi,
                                                uint32_t p2lo, uint32_t p2hi) {
  const u16x2 v = as_pk(v2), a = as_pk(a2);
  const u16x2 mn = __builtin_elementwise_min(v, a), mx = __builtin_elementwise_max(v, a);
  const u16x2 t = mx - mn;
  const u16x2 c8 = {8, 8}, c12 = {12, 12}, c16 = {16, 16};
  // selectors: byte 0 / 2 = index into the piece (12 = "zero" for v_perm_b32), bytes 1 / 3 = 12
  const uint32_t hi12 = 0x0C000C00u;
  const uint32_t s0 = as_u32(__builtin_elementwise_min(t, c12)) | hi12;
  const uint32_t s1 = as_u32(__builtin_elementwise_min(__builtin_elementwise_sub_sat(t, c8), c12)) | hi12;
  const uint32_t s2 = as_u32(__builtin_elementwise_min(__builtin_elementwise_sub_sat(t, c16), c12)) | hi12;
  const uint32_t l0 = __builtin_amdgcn_perm(p0hi, p0lo, s0), l1 = __builtin_amdgcn_perm(p1hi, p1lo, s1), l2 = __builtin_amdgcn_perm(p2hi, p2lo, s2);
  const u16x2 tab = __builtin_elementwise_max(__builtin_elementwise_max(as_pk(l0), as_pk(l1)), as_pk(l2));
  return as_u32(__builtin_elementwise_sub_sat(mn, tab));
}

template <int FORM>
__global__ __launch_bounds__(256) void fold_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t d, uint32_t rounds) {
  extern __shared__ uint32_t lds[];
  uint32_t *col = lds + threadIdx.x;                     // [slot][thread]
  uint8_t *tab2d = reinterpret_cast<uint8_t *>(lds + 32 * 256);  // form 2: [acc 0..127][v 0..127]
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t j = 0; j < d; j++) col[j * 256] = in[(size_t(gid) * 32 + j)] & 0x7F7F7F7Fu;
  if (FORM == 2) {
    for (uint32_t k = threadIdx.x; k < 128 * 128; k += blockDim.x) {
      const uint32_t a = k >> 7, v = k & 127u;
      const uint32_t t = a > v ? a - v : v - a;
      const int m = int(min(a, v)) - table_entry(t);
      tab2d[k] = static_cast<uint8_t>(m < 0 ? 0 : m);
    }
  }
  __syncthreads();
  // the table pieces of form 1 (bytes: T[0..7], T[8..15], T[16..23])
  uint32_t p[6];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    uint32_t w = 0;
    for (int b = 0; b < 4; b++) w |= uint32_t(table_entry(q * 4 + b)) << (8 * b);
    p[q] = w;
  }
  uint32_t acc[4] = {127u, 127u, 127u, 127u};
  uint32_t alo = 0x007F007Fu, ahi = 0x007F007Fu;
  for (uint32_t r = 0; r < rounds; r++) {
    for (uint32_t j = 0; j < d; j++) {
      const uint32_t mj = col[j * 256];
      if (FORM == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = step_popcount((mj >> (8 * k)) & 0xFFu, acc[k]);
      } else if (FORM == 1) {
        alo = step_packed(mj & 0x00FF00FFu, alo, p[0], p[1], p[2], p[3], p[4], p[5]);
        ahi = step_packed((mj >> 8) & 0x00FF00FFu, ahi, p[0], p[1], p[2], p[3], p[4], p[5]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = tab2d[(acc[k] << 7) | ((mj >> (8 * k)) & 0xFFu)];
      }
    }
    // keep the chain alive across rounds without letting it settle at zero: re-seed from the column
    if (FORM == 1) { alo |= col[(r % d) * 256] & 0x00400040u; ahi |= (col[(r % d) * 256] >> 8) & 0x00400040u; }
    else {
#pragma unroll
      for (int k = 0; k < 4; k++) acc[k] |= (col[(r % d) * 256] >> (8 * k)) & 0x40u;
    }
  }
  uint32_t res = FORM == 1 ? ((alo & 0xFFu) | ((ahi & 0xFFu) << 8) | ((alo >> 16) << 16) | ((ahi >> 16) << 24))
                           : (acc[0] | (acc[1] << 8) | (acc[2] << 16) | (acc[3] << 24));
  out[gid] = res;
}

int main(int argc, char **argv) {
  uint32_t d = 7, rounds = 2000, blocks = 256 * 16;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--d")) d = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rounds")) rounds = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--blocks")) blocks = atoi(argv[++i]);
  }
  const size_t threads = size_t(blocks) * 256;
  std::vector<uint32_t> h(threads * 32);
  uint64_t x = 88172645463325252ull;
  for (auto &w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = static_cast<uint32_t>(x >> 16); }
  uint32_t *in, *out;
  CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&out, threads * 4));
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t lds0 = 32 * 256 * 4, lds2 = lds0 + 128 * 128;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fold_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds2)));
  std::vector<uint32_t> res(threads);
  for (int form = 0; form < 3; form++) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(a));
      if (form == 0) fold_kernel<0><<<blocks, 256, lds0>>>(in, out, d, rounds);
      else if (form == 1) fold_kernel<1><<<blocks, 256, lds0>>>(in, out, d, rounds);
      else fold_kernel<2><<<blocks, 256, lds2>>>(in, out, d, rounds);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep > 0 && ms < best) best = ms;
    }
    CK(hipMemcpy(res.data(), out, threads * 4, hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (auto w : res) sum = sum * 1000003ull + w;
    const double steps = double(threads) * rounds * d * 4;
    printf("form %d (%s), d = %u: %.3f ms for %.3g byte-steps = %.1f us per 1e9 steps  (checksum %016llx)\n", form,
           form == 0 ? "popcount table, today's" : (form == 1 ? "packed 16-bit + v_perm_b32" : "16 KB [acc][v] table in LDS"), d, best, steps,
           best * 1e3 / (steps / 1e9), (unsigned long long)sum);
  }
  return 0;
}
