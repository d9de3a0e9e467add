// Division by a launch-invariant divisor as a multiplication: host side makes the constants, kernels use them.
// Plain C++ (no HIP needed): tests/test_host_logic.py compiles this header with g++ and checks it against `/`.
#pragma once
#include <cstdint>

#ifdef __HIPCC__
#define LDPC_FD_INLINE __forceinline__
#else
#define LDPC_FD_INLINE inline
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif

namespace ldpc {
namespace dev {

// Division by a launch-invariant divisor as a multiplication (round 5).  A division by a run-time value costs the
// scalar unit about twenty dependent instructions plus five vector ones (the compiler goes through v_rcp_iflag_f32
// and two correction steps); a wavefront of the per-level launches handles ONE check row, and the five divisions of its
// prologue (wave -> tile / slice / node, slice -> tile base) were 159 of the ~425 scalar instructions it executes
// (profiles/r05_config3_salu.txt).  q = (n * mul) >> shr is exact for every n < 2^31 with shr = 31 + ceil(log2 d),
// mul = ceil(2^shr / d) (the error mul * d - 2^shr is below d <= 2^(shr - 31), so n * error < 2^shr).
struct FastDiv {
  uint32_t d, mul, shr;
};
__host__ inline FastDiv fast_div(uint32_t d) {
  FastDiv f{d ? d : 1u, 0, 31};
  while ((uint64_t(1) << (f.shr - 31)) < f.d) f.shr++;
  f.mul = static_cast<uint32_t>(((uint64_t(1) << f.shr) + f.d - 1) / f.d);  // d = 1: 2^31; d = 2^s: 2^31 too
  return f;
}
__host__ __device__ LDPC_FD_INLINE uint32_t fdiv_q(uint32_t n, const FastDiv &f) {
  return static_cast<uint32_t>((uint64_t(n) * f.mul) >> f.shr);
}

}  // namespace dev
}  // namespace ldpc
