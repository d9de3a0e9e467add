// Standalone reproducer for round 4's "element 2 of lanes 12-15 of every 16 differs from run to run" in cn_minsum_rec_kernel.
// Hypothesis (from the ISA of the failing build, tools/mb/store_hazard_scan.py): on gfx950 a 128-bit MUBUF store whose soffset is an
// SGPR still reads its data registers AFTER issue, so a VALU instruction in the next slot that rewrites one of them corrupts the
// store.  The GCN3/Vega manuals list this hazard (">64-bit VMEM store, then VALU write of vdata: 1 wait state") but exempt MUBUF
// stores with an SGPR soffset, and LLVM's GCNHazardRecognizer::createsVALUHazard follows the exemption -- which is exactly the form
// buf_store emits (`buffer_store_dwordx4 v[a:d], v, s[r:r+3], sN offen`).
//   hipcc -O2 --offload-arch=gfx950 tools/mb/store_hazard_repro.hip -o tools/mb/store_hazard_repro && tools/mb/store_hazard_repro
// Every wave stores {tag0..tag3} per lane with one dwordx4 store and rewrites data register ELT with 0xdeadbeef NOPS wait states
// later; the host counts poisoned words by (lane % 16, element).  A correct machine (or enough wait states) reports 0 everywhere.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kPoison = 0xdeadbeefu, kRows = 64;

// SOFF: 1 = soffset in an SGPR (the library's form, the one the compiler never pads), 0 = soffset is the literal 0 (the form the
// hazard recogniser pads with wait states when the compiler itself schedules the code; here nothing is padded: inline asm)
template <int NOPS, int ELT, int SOFF, int NT>
__global__ __launch_bounds__(256) void store_then_clobber(uint32_t *out, uint32_t row_bytes) {
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t *base = out + size_t(wave) * kRows * (row_bytes / 4);
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(a >> 32)))) << 32) |
                     uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(uint32_t(a))));
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(u), 0, int(kRows * row_bytes), 0x00020000);
  for (uint32_t row = 0; row < kRows; row++) {
    const uint32_t tag = (wave << 16) | (row << 8) | (lane << 2);
    const uint32_t voff = SOFF ? lane * 16u : lane * 16u + row * row_bytes;
    const uint32_t soff = __builtin_amdgcn_readfirstlane(row * row_bytes);
    asm volatile(
        "v_mov_b32 v20, %0\n v_or_b32 v21, 1, %0\n v_or_b32 v22, 2, %0\n v_or_b32 v23, 3, %0\n s_nop 4\n"
        "%=:\n"
        ".if %5\n .if %6\n buffer_store_dwordx4 v[20:23], %1, %2, %3 offen nt\n .else\n buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n .endif\n"
        ".else\n .if %6\n buffer_store_dwordx4 v[20:23], %1, %2, 0 offen nt\n .else\n buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n .endif\n .endif\n"
        ".rept %4\n s_nop 0\n .endr\n"
        ".if %7 == 0\n v_mov_b32 v20, 0xdeadbeef\n .endif\n .if %7 == 1\n v_mov_b32 v21, 0xdeadbeef\n .endif\n"
        ".if %7 == 2\n v_mov_b32 v22, 0xdeadbeef\n .endif\n .if %7 == 3\n v_mov_b32 v23, 0xdeadbeef\n .endif\n"
        :: "v"(tag), "v"(voff), "s"(rsrc), "s"(soff), "n"(NOPS), "n"(SOFF), "n"(NT), "n"(ELT)
        : "v20", "v21", "v22", "v23", "memory");
  }
}

template <int NOPS, int ELT, int SOFF, int NT>
static int run(uint32_t *d, std::vector<uint32_t> &h, uint32_t waves, uint32_t row_bytes) {
  HIP_OK(hipMemset(d, 0, h.size() * 4));
  hipLaunchKernelGGL((store_then_clobber<NOPS, ELT, SOFF, NT>), dim3(waves / 4), dim3(256), 0, 0, d, row_bytes);
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  uint64_t bad[16][4] = {}, wrong = 0, total = 0;
  for (size_t i = 0; i < h.size(); i++) {
    const uint32_t lane = (i / 4) % 64, e = i % 4, row = (i / 256) % kRows, wave = i / (256 * kRows);
    const uint32_t want = ((wave << 16) | (row << 8) | (lane << 2)) | e;
    if (h[i] == kPoison) { bad[lane % 16][e]++; total++; } else if (h[i] != want) wrong++;
  }
  printf("soffset=%s nt=%d clobbered element %d, %d wait state(s): %8llu poisoned of %zu words, %llu otherwise wrong", SOFF ? "SGPR" : "0   ",
         NT, ELT, NOPS, (unsigned long long)total, h.size() / 4, (unsigned long long)wrong);
  if (total) {
    printf("   [lane%%16: count]");
    for (int l = 0; l < 16; l++) if (bad[l][ELT]) printf(" %d:%llu", l, (unsigned long long)bad[l][ELT]);
  }
  printf("\n");
  return total != 0;
}

int main() {
  const uint32_t waves = 256 * 32, row_bytes = 1024;   // 8192 waves x 64 rows x 1 KiB = 512 MiB: the stores queue up behind HBM
  std::vector<uint32_t> h(size_t(waves) * kRows * (row_bytes / 4));
  uint32_t *d;
  HIP_OK(hipMalloc(&d, h.size() * 4));
  int hits = 0;
  for (int rep = 0; rep < 2; rep++) {
#define ROW(S, N) hits += run<0, 0, S, N>(d, h, waves, row_bytes) + run<0, 1, S, N>(d, h, waves, row_bytes) + run<0, 2, S, N>(d, h, waves, row_bytes) + \
    run<0, 3, S, N>(d, h, waves, row_bytes) + run<1, 2, S, N>(d, h, waves, row_bytes) + run<1, 3, S, N>(d, h, waves, row_bytes) +                       \
    run<2, 2, S, N>(d, h, waves, row_bytes) + run<2, 3, S, N>(d, h, waves, row_bytes) + run<3, 3, S, N>(d, h, waves, row_bytes);
    ROW(1, 1) ROW(1, 0) ROW(0, 1) ROW(0, 0)
  }
  printf("%s\n", hits ? "HAZARD REPRODUCED: a 128-bit MUBUF store reads its data after the next VALU instruction has issued" : "no corruption seen");
  return 0;
}
