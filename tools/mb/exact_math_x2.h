// Two-arguments-per-lane forms of exact_math.h's tanhf_c9 / log1pf / atanh_rs, written to find out whether CDNA4's
// packed v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 buy anything for the sum-product rules.  NOT part of the product:
// they are bit-identical to glibc in both positions on all 2^32 arguments (checked on gfx950, round 3), need a quarter
// fewer vector instructions per evaluation (126 against 172) -- and are SLOWER (302 against 347 G evaluations/s),
// because on gfx950 a wave64 v_fma_f32 already issues every ~2.4 cycles and a v_pk_fma_f32 takes twice that: packing
// saves instructions, not issue cycles (profiles/r03_packed_f32.txt; tools/mb/pk_bench.hip).
#pragma once
#include "../../ldpc_toolbox_amd/csrc/exact_math.h"

namespace ldpc {
namespace em {

// ---------------------------------------------------------------------------------------------
// TWO arguments per lane (HIP compilations only): the same operation sequences as tanhf_c9 / log1pf / atanh_rs above on
// 2-vectors, so that the float multiplies, adds and fused multiply-adds -- about half of each function's
// instructions -- issue as CDNA's packed v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (two IEEE operations per lane
// per instruction, same rounding as the scalar forms).  The integer steps, conversions, reciprocals and selects
// stay per element.  The kernels that are bound by vector-ALU issue (the Tanh rule) give a lane two codewords and
// call these.  Per element the operations are those of the scalar functions; checked against glibc on every
// float, in both positions, on the device (tools/check_exact_math_device.hip).
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__) || defined(__HIP__)  // (clang vector types; the host pass of a HIP compilation sees them too)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
EM_FN u32x2 as_u32(f32x2 x) { return __builtin_bit_cast(u32x2, x); }
EM_FN f32x2 as_f32(u32x2 x) { return __builtin_bit_cast(f32x2, x); }
EM_FN f32x2 splat2(float a) { return f32x2{a, a}; }
// m: all ones / zero per element (the result type of a vector comparison)
EM_FN f32x2 sel2(i32x2 m, f32x2 a, f32x2 b) {
  return as_f32((as_u32(a) & __builtin_bit_cast(u32x2, m)) | (as_u32(b) & ~__builtin_bit_cast(u32x2, m)));
}
EM_FN u32x2 sel2(i32x2 m, u32x2 a, u32x2 b) {
  return (a & __builtin_bit_cast(u32x2, m)) | (b & ~__builtin_bit_cast(u32x2, m));
}
EM_FN i32x2 sel2(i32x2 m, i32x2 a, i32x2 b) { return (a & m) | (b & ~m); }
EM_FN f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// fdiv_v above, two quotients (gfx950 only: see there)
template <int V>
EM_FN f32x2 fdiv_v(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__)
  f32x2 r = f32x2{__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
  if (V == 0 || V == 1) {
    const f32x2 e0 = fma2(-b, r, splat2(1.0f));
    r = fma2(e0, r, r);
  }
  f32x2 q = a * r;
  if (V == 4) return q;
  const f32x2 e1 = fma2(-b, q, a);
  q = fma2(e1, r, q);
  if (V == 0 || V == 3) {
    const f32x2 e2 = fma2(-b, q, a);
    q = fma2(e2, r, q);
  }
  return q;
#else
  return a / b;
#endif
}

EM_FN f32x2 tanhf_c9(f32x2 x) {
  const f32x2 one = splat2(1.0f), ln2_hi = splat2(6.9313812256e-01f), ln2_lo = splat2(9.0580006145e-06f),
              invln2 = splat2(1.4426950216e+00f);
  const f32x2 Q1 = splat2(-3.3333335072e-02f), Q2 = splat2(1.5873016091e-03f), Q3 = splat2(-7.9365076090e-05f),
              Q4 = splat2(4.0082177293e-06f), Q5 = splat2(-2.0109921195e-07f);
  const u32x2 jx = as_u32(x);
  const u32x2 ix = jx & 0x7fffffffu;
  const f32x2 ax = as_f32(ix);
  const i32x2 big = ix >= 0x3f800000u;
  const f32x2 arg = ax * sel2(big, splat2(2.0f), splat2(-2.0f));
  const u32x2 hx = as_u32(arg) & 0x7fffffffu;
  const f32x2 g = __builtin_convertvector(__builtin_convertvector(invln2 * arg + sel2(big, splat2(0.5f), splat2(-0.5f)), i32x2), f32x2);
  f32x2 kf = sel2(hx < 0x3F851592u, splat2(-1.0f), g);
  kf = sel2(hx > 0x3eb17218u, kf, splat2(0.0f));
  const i32x2 k = __builtin_convertvector(kf, i32x2);
  const f32x2 hi = arg - kf * ln2_hi;
  const f32x2 lo = kf * ln2_lo;
  const f32x2 xr = hi - lo;
  const f32x2 c = (hi - xr) - lo;
  const f32x2 hfx = 0.5f * xr;
  const f32x2 hxs = xr * hfx;
  const f32x2 r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  const f32x2 t3 = 3.0f - r1 * hfx;
  const f32x2 e = hxs * fdiv_v<EM_FDIV_EXPM1>(r1 - t3, 6.0f - xr * t3);
  const f32x2 r0 = xr - (xr * e - hxs);
  const f32x2 e2 = (xr * (e - c) - c) - hxs;
  const f32x2 rm1 = 0.5f * (xr - e2) - 0.5f;
  const f32x2 dd = e2 - xr;
  const u32x2 ku = __builtin_bit_cast(u32x2, k);
  const u32x2 kbits = ku << 23;
  const f32x2 ya = as_f32(as_u32(one - dd) + kbits) - one;
  const u32x2 ksmall = sel2((k >= 2) & (k < 23), ku, u32x2{2u, 2u});
  const f32x2 t1 = as_f32(0x3f800000u - (u32x2{0x1000000u, 0x1000000u} >> ksmall));
  const f32x2 yb = as_f32(as_u32(t1 - dd) + kbits);
  const u32x2 klarge = sel2(k >= 23, ku, u32x2{23u, 23u});
  const f32x2 t2 = as_f32((0x7fu - klarge) << 23);
  const f32x2 yc = as_f32(as_u32((xr - (e2 + t2)) + one) + kbits);
  const f32x2 r_big = sel2(k < 23, yb, yc);
  const f32x2 r_small = sel2(k == 0, r0, sel2(k == -1, rm1, ya));
  f32x2 t = sel2(big, r_big, r_small);
  t = sel2(hx < 0x33000000u, arg, t);
  const f32x2 q = fdiv_v<EM_FDIV_TANH>(sel2(big, splat2(2.0f), -t), t + 2.0f);
  const f32x2 z = sel2(big, one - q, q);
  return as_f32(as_u32(z) | (jx & 0x80000000u));
}

// (the rare classes of either element: the scalar functions, out of line -- inlined twice per call site they
// tripled the callers' code and cost registers)
#if defined(__HIP_DEVICE_COMPILE__)
#define EM_SLOW __device__ __attribute__((noinline))
#else
#define EM_SLOW static __attribute__((noinline))
#endif
EM_SLOW f32x2 log1pf_pair_slow(f32x2 x) { return f32x2{log1pf(x.x), log1pf(x.y)}; }
EM_SLOW f32x2 atanh_pair_slow(f32x2 x) { return f32x2{atanh_rs(x.x), atanh_rs(x.y)}; }

EM_FN f32x2 log1pf(f32x2 x) {
  const f32x2 ln2_hi = splat2(6.9313812256e-01f), ln2_lo = splat2(9.0580006145e-06f);
  const f32x2 Lp1 = splat2(6.6666668653e-01f), Lp2 = splat2(4.0000000596e-01f), Lp3 = splat2(2.8571429849e-01f),
              Lp4 = splat2(2.2222198546e-01f), Lp5 = splat2(1.8183572590e-01f), Lp6 = splat2(1.5313838422e-01f),
              Lp7 = splat2(1.4798198640e-01f);
  const i32x2 hx = __builtin_bit_cast(i32x2, x);
  const i32x2 ax = hx & 0x7fffffff;
  const i32x2 special = ((hx < 0) & (ax >= 0x3f800000)) | (ax < 0x31000000) | (hx >= 0x5a000000);
  const i32x2 direct = (hx < 0x3ed413d7) & ((hx > 0) | (hx <= static_cast<int32_t>(0xbe95f61f)));
  const f32x2 u0 = 1.0f + x;
  i32x2 hu = __builtin_bit_cast(i32x2, u0);
  i32x2 k = (hu >> 23) - 127;
  f32x2 c = sel2(k > 0, 1.0f - (u0 - x), x - (u0 - 1.0f));
  c = fdiv_v<EM_FDIV_L1P_C>(c, u0);
  hu &= 0x007fffff;
  const i32x2 low = hu < 0x3504f7;
  k += sel2(low, i32x2{0, 0}, i32x2{1, 1});
  const f32x2 u = __builtin_bit_cast(f32x2, hu | sel2(low, i32x2{0x3f800000, 0x3f800000}, i32x2{0x3f000000, 0x3f000000}));
  hu = sel2(low, hu, (0x00800000 - hu) >> 2);
  const i32x2 rare = special | (~direct & (hu == 0));
  if (rare.x | rare.y) return log1pf_pair_slow(x);  // the scalar function's own classes
  const f32x2 f = sel2(direct, x, u - 1.0f);
  k = sel2(direct, i32x2{0, 0}, k);
  c = sel2(direct, splat2(0.0f), c);
  const f32x2 hfsq = 0.5f * f * f;
  const f32x2 s = fdiv_v<EM_FDIV_L1P_S>(f, 2.0f + f);
  const f32x2 z = s * s;
  const f32x2 R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  const f32x2 kf = __builtin_convertvector(k, f32x2);
  const f32x2 r0 = f - (hfsq - s * (hfsq + R));
  const f32x2 rk = kf * ln2_hi - ((hfsq - (s * (hfsq + R) + (kf * ln2_lo + c))) - f);
  return sel2(k == 0, r0, rk);
}

EM_FN f32x2 atanh_rs(f32x2 x) {
  if (!(__builtin_fabsf(x.x) < 1.0f && __builtin_fabsf(x.y) < 1.0f)) return atanh_pair_slow(x);
  const f32x2 q = fdiv_v<EM_FDIV_ATANH>(2.0f * x, 1.0f - x);
  return 0.5f * log1pf(sel2(x == 0.0f, x, q));
}
#endif  // HIP compilation

}  // namespace em
}  // namespace ldpc
