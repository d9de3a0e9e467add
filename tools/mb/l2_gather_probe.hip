// Probe for the one structural lever left on the headline kernel (round 5 review, item 3): would the check-node launch's posterior
// gathers run faster from an XCD's L2 than from the Infinity Cache?  Today a tile is 256 codewords: a gathered posterior row is
// 1 KiB, a tile's gathered rows are 33 MB (Infinity Cache).  With 16-codeword tiles pinned to XCDs a tile's gathered rows are
// 2 MB (L2), a row is 64 B and a quarter-wave owns a check row.  Modes (same loop, --mode):
//   0  1-KiB row segments (float4 per lane) of ONE 33 MB table, uniformly random rows           -> Infinity Cache gathers alone
//   1  64-B row segments (one dword per lane, a row per quarter-wave) of a 2 MB table per XCD    -> L2 gathers alone
//   2  mode 1 + the check-node kernel's streams per 5 gathers: 6 nontemporal loads, 6 nontemporal stores of the same width
//   3  mode 0 + the same streams (float4)                                                        -> today's kernel in miniature
//   4  streams only, the variable-node kernel's mix: 5 nontemporal float4 loads + 1 plain float4 store per step, no gathers
//   5  streams only: 6 loads + 6 stores (a copy)
// Output per mode: gathered TB/s, streamed TB/s, codeword-rows per second (the figure that decides).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ table, const float *__restrict__ sin, float *__restrict__ sout,
                                             float *__restrict__ sink, uint32_t rows, uint32_t iters, uint32_t *xcc_mismatch) {
  constexpr bool WIDE = MODE == 0 || MODE >= 3, STREAMS = MODE >= 2, GATHER = MODE <= 3;
  constexpr int NST = MODE == 4 ? 1 : 6, NLD = MODE == 4 ? 5 : 6;
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t xcd = blockIdx.x & 7u;
  if (threadIdx.x == 0) {
    const uint32_t id = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID
    if (id != xcd) atomicAdd(xcc_mismatch, 1u);
  }
  const size_t per_xcd = size_t(rows) * 16;  // floats of one XCD's 64-B-row table
  const float *tab = WIDE ? table : table + xcd * per_xcd;
  // a wave's private stream region: [iters][6] wave-instructions in, the same out
  const size_t words = WIDE ? 256 : 64;
  const float *in = sin + size_t(wave) * iters * 6 * words + (WIDE ? lane * 4 : lane);
  float *out = sout + size_t(wave) * iters * 6 * words + (WIDE ? lane * 4 : lane);
  float acc = 0.f;
  for (uint32_t it = 0; it < iters; it++) {
    f4 g[5];
    f4 s[6];
#pragma unroll
    for (int j = 0; j < (GATHER ? 5 : 0); j++) {
      if (WIDE) {
        const uint32_t r = mix(wave * 9781u + it * 5u + j) % rows;
        g[j] = *reinterpret_cast<const f4 *>(tab + size_t(r) * 256 + lane * 4);
      } else {
        const uint32_t r = mix((wave * 4u + (lane >> 4)) * 9781u + it * 5u + j) % rows;  // a row per quarter-wave
        g[j].x = tab[size_t(r) * 16 + (lane & 15u)];
      }
    }
    if (STREAMS) {
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const float *p = in + (size_t(it) * 6 + j) * words;
        if (WIDE) s[j] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
        else s[j].x = __builtin_nontemporal_load(p);
      }
    }
#pragma unroll
    for (int j = 0; j < (GATHER ? 5 : 0); j++) acc += WIDE ? (g[j].x + g[j].y + g[j].z + g[j].w) : g[j].x;
    if (!GATHER) {
#pragma unroll
      for (int j = 1; j < NLD; j++) s[0] += s[j];
    }
    if (STREAMS) {
#pragma unroll
      for (int j = 0; j < NST; j++) {
        float *p = out + (size_t(it) * 6 + j) * words;
        if (WIDE && MODE == 4) { f4 v = s[j]; v.x += acc; *reinterpret_cast<f4 *>(p) = v; }
        else if (WIDE) { f4 v = s[j]; v.x += acc; __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p)); }
        else __builtin_nontemporal_store(s[j].x + acc, p);
      }
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char **argv) {
  int mode = 0, reps = 5;
  uint32_t rows = 32400, iters = 0, blocks = 4096;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--mode")) mode = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rows")) rows = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--blocks")) blocks = atoi(argv[++i]);
  }
  const bool wide = mode == 0 || mode >= 3, streams = mode >= 2;
  if (!iters) iters = wide ? 64 : 256;
  const size_t waves = size_t(blocks) * 4, words = wide ? 256 : 64;
  const size_t table_floats = wide ? size_t(rows) * 256 : size_t(rows) * 16 * 8;
  const size_t stream_floats = streams ? waves * iters * 6 * words : 64;
  float *table, *sin, *sout, *sink;
  uint32_t *mism;
  CK(hipMalloc(&table, table_floats * 4)); CK(hipMalloc(&sin, stream_floats * 4)); CK(hipMalloc(&sout, stream_floats * 4));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&mism, 4));
  CK(hipMemset(table, 0, table_floats * 4)); CK(hipMemset(sin, 0, stream_floats * 4)); CK(hipMemset(mism, 0, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; r++) {
    CK(hipEventRecord(a));
    switch (mode) {
      case 0: probe<0><<<blocks, 256>>>(table, sin, sout, sink, rows, iters, mism); break;
      case 1: probe<1><<<blocks, 256>>>(table, sin, sout, sink, rows, iters, mism); break;
      case 2: probe<2><<<blocks, 256>>>(table, sin, sout, sink, rows, iters, mism); break;
      case 3: probe<3><<<blocks, 256>>>(table, sin, sout, sink, rows, iters, mism); break;
      case 4: probe<4><<<blocks, 256>>>(table, sin, sout, sink, rows, iters, mism); break;
      default: probe<5><<<blocks, 256>>>(table, sin, sout, sink, rows, iters, mism); break;
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0 && ms < best) best = ms;
  }
  uint32_t mm; CK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
  const double s = best * 1e-3, gather_bytes = mode <= 3 ? double(waves) * iters * 5 * words * 4 : 0, stream_bytes = streams ? double(waves) * iters * (mode == 4 ? 6 : 12) * words * 4 : 0;
  const double cw_rows = double(waves) * iters * (wide ? 256 : 64);  // codeword-rows: a row's 5 gathers for 256 (wide) / 4 x 16 codewords
  printf("mode %d: %s, table %.1f MB%s, %u blocks x %u iters: %.3f ms  gathers %.2f TB/s  streams %.2f TB/s  total %.2f TB/s  %.1f G codeword-rows/s  (workgroups off their blockIdx%%8 XCD: %u of %u)\n",
         mode, wide ? "1-KiB segments" : "64-B segments (row per quarter-wave)", (wide ? table_floats : table_floats / 8) * 4 / 1e6,
         wide ? "" : " per XCD", blocks, iters, best, gather_bytes / s / 1e12, stream_bytes / s / 1e12, (gather_bytes + stream_bytes) / s / 1e12,
         cw_rows / s / 1e9, mm, blocks * (reps + 1));
  return 0;
}
