// exp / log / log1p / tanh (single and double precision) that return, bit for bit, what glibc
// 2.35's libm returns on x86-64 -- the functions Rust's f32/f64::{exp, ln, ln_1p, tanh} resolve to
// on linux-gnu, i.e. the ones the reference's transcendental decoder rules are built on
// (/root/reference/src/decoder/arithmetic.rs:184, 357, 376, 510, 965-966).  ocml's versions
// differ in the last ulp, which the ill-conditioned rules (phi's clamp cliff, A-Min*'s argmin
// ties) occasionally amplify; with these the HIP path reproduces the CPU results exactly.
//
// The algorithms are the published ones glibc uses:
//   expf, logf      Szabolcs Nagy's table-driven routines (ARM optimized-routines), evaluated
//                   in double precision; the x86-64 build that runs on FMA-capable CPUs fuses
//                   specific multiply-adds, reproduced here with explicit fma() calls
//   log1pf, expm1f, tanhf   the Sun fdlibm float routines
// Every constant was checked against the libm binary; the f32 functions are compared with the
// host libm exhaustively over all 2^32 arguments (tools/check_exact_math.cpp), the f64 ones on
// 2e9 sampled arguments each (tools/check_exact_math64.cpp); tests/test_exact_math.py re-checks a
// sample in the CPU suite.
// Works as host code (for that test) and as HIP device code; needs -ffp-contract=off.
//
// Provenance and licences of the algorithms re-implemented here (no source text was copied; the
// operation sequences and constants are those of the published routines, which is what makes the
// results bit-identical):
//   * expf, logf, exp, log: Szabolcs Nagy, ARM Optimized Routines (math/expf.c, logf.c, exp.c, log.c),
//     as adopted by glibc 2.28+ (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, dbl-64/e_exp.c, e_log.c).
//     Copyright (c) Arm Limited; MIT licence (ARM Optimized Routines) / LGPL-2.1-or-later (the glibc copy).
//   * log1pf, expm1f, tanhf, log1p, expm1, tanh: FreeBSD msun / Sun fdlibm (s_log1pf.c, s_expm1f.c,
//     s_tanhf.c and the double versions), as shipped in glibc's sysdeps/ieee754.
//     "Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunPro, a Sun
//     Microsystems, Inc. business.  Permission to use, copy, modify, and distribute this software is
//     freely granted, provided that this notice is preserved."
//   * atanh: the one-line formula of Rust's standard library (library/std/src/f32.rs; MIT OR Apache-2.0).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define EM_FN __host__ __device__ __forceinline__
#else
#define EM_FN static inline
#endif

namespace ldpc {
namespace em {

EM_FN uint32_t as_u32(float x) { return __builtin_bit_cast(uint32_t, x); }
EM_FN float as_f32(uint32_t x) { return __builtin_bit_cast(float, x); }
EM_FN uint64_t as_u64(double x) { return __builtin_bit_cast(uint64_t, x); }
EM_FN double as_f64(uint64_t x) { return __builtin_bit_cast(double, x); }

// 2^(i/32) tables of exp2f_data: T[i] = bits(2^(i/32)) - (i << 47)
EM_FN uint64_t exp2f_tab(uint32_t i) {
  constexpr uint64_t T[32] = {
      0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
      0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
      0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
      0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
      0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
      0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
      0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
      0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
  return T[i];
}

// a / b for the fast paths below.  On the device: reciprocal-refinement sequences built on v_rcp_f32
// (1 ulp) without the operand pre-scaling and special-case fix-up of the compiler's division
// (v_div_scale / v_div_fmas / v_div_fixup) -- the callers' operands are finite, non-zero divisors away
// from the exponent limits.  Each call site names the shortest sequence that leaves its FUNCTION equal to
// glibc on every one of its 2^32 arguments (tools/check_exact_math_device.hip; a division inside a
// function of one argument only ever sees the operand pairs that argument produces, so the exhaustive
// check of the function is an exhaustive check of the site):
//   V = 0   r refined once, quotient corrected twice (8 instructions): correctly rounded in general
//   V = 1   r refined once, quotient corrected once (6)
//   V = 2   quotient corrected once with the raw reciprocal (4)
//   V = 3   quotient corrected twice with the raw reciprocal (6)
//   V = 4   a * rcp(b), uncorrected (2): only where the quotient is a small correction term
// The exhaustive checks that license the shortened sequences ran on gfx950 -- the rounding of ITS v_rcp_f32 is
// part of the proof -- so only a gfx950 device build takes them; any other device target, and the host, use the
// correctly rounded IEEE division (slower, exact everywhere).
template <int V>
EM_FN float fdiv_v(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__)
  float r = __builtin_amdgcn_rcpf(b);
  if (V == 0 || V == 1) {
    const float e0 = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
  }
  float q = a * r;
  if (V == 4) return q;
  const float e1 = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(e1, r, q);
  if (V == 0 || V == 3) {
    const float e2 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e2, r, q);
  }
  return q;
#else
  return a / b;
#endif
}
// per-site choice (overridable on the command line of the check tool to search for the shortest)
#ifndef EM_FDIV_EXPM1
#define EM_FDIV_EXPM1 2
#endif
#ifndef EM_FDIV_TANH
#define EM_FDIV_TANH 2
#endif
#ifndef EM_FDIV_L1P_C
#define EM_FDIV_L1P_C 4
#endif
#ifndef EM_FDIV_L1P_S
#define EM_FDIV_L1P_S 2
#endif
#ifndef EM_FDIV_ATANH
#define EM_FDIV_ATANH 2
#endif
EM_FN float fdiv(float a, float b) { return fdiv_v<0>(a, b); }

// glibc 2.35 sysdeps/ieee754/flt-32/e_expf.c, FMA build
EM_FN float expf(float x) {
  const double xd = static_cast<double>(x);
  const uint32_t ix = as_u32(x);
  const uint32_t abstop = (ix >> 20) & 0x7ff;
  if (abstop >= 0x42b) {  // |x| >= 88 or NaN
    if (ix == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return 0x1p97f * 0x1p97f;       // overflow
    if (x < -0x1.9fe368p6f) return 0x1p-95f * 0x1p-95f;    // underflow
    if (x < -0x1.9d1d9ep6f) return 0x1.4p-75f * 0x1.4p-75f;  // may underflow
  }
  const double invln2n = 0x1.71547652b82fep+5, shift = 0x1.8p+52;
  double kd = __builtin_fma(invln2n, xd, shift);
  const uint64_t ki = as_u64(kd);
  kd -= shift;
  const double r = __builtin_fma(invln2n, xd, -kd);
  const uint64_t t = exp2f_tab(static_cast<uint32_t>(ki & 31)) + (ki << 47);
  const double s = as_f64(t);
  const double z = __builtin_fma(r, 0x1.c6af84b912394p-20, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double y = __builtin_fma(r, 0x1.62e42ff0c52d6p-6, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return static_cast<float>(y);
}

EM_FN void logf_tab(uint32_t i, double *invc, double *logc) {
  constexpr uint64_t T[32] = {
      0x3ff661ec79f8f3beULL, 0xbfd57bf7808caadeULL, 0x3ff571ed4aaf883dULL, 0xbfd2bef0a7c06ddbULL,
      0x3ff49539f0f010b0ULL, 0xbfd01eae7f513a67ULL, 0x3ff3c995b0b80385ULL, 0xbfcb31d8a68224e9ULL,
      0x3ff30d190c8864a5ULL, 0xbfc6574f0ac07758ULL, 0x3ff25e227b0b8ea0ULL, 0xbfc1aa2bc79c8100ULL,
      0x3ff1bb4a4a1a343fULL, 0xbfba4e76ce8c0e5eULL, 0x3ff12358f08ae5baULL, 0xbfb1973c5a611cccULL,
      0x3ff0953f419900a7ULL, 0xbfa252f438e10c1eULL, 0x3ff0000000000000ULL, 0x0000000000000000ULL,
      0x3fee608cfd9a47acULL, 0x3faaa5aa5df25984ULL, 0x3feca4b31f026aa0ULL, 0x3fbc5e53aa362eb4ULL,
      0x3feb2036576afce6ULL, 0x3fc526e57720db08ULL, 0x3fe9c2d163a1aa2dULL, 0x3fcbc2860d224770ULL,
      0x3fe886e6037841edULL, 0x3fd1058bc8a07ee1ULL, 0x3fe767dcf5534862ULL, 0x3fd4043057b6ee09ULL};
  *invc = as_f64(T[2 * i]);
  *logc = as_f64(T[2 * i + 1]);
}

// glibc 2.35 sysdeps/ieee754/flt-32/e_logf.c, FMA build
EM_FN float logf(float x) {
  uint32_t ix = as_u32(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return -1.0f / 0.0f;                   // log(+-0) = -inf
    if (ix == 0x7f800000u) return x;                        // log(inf) = inf
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);  // invalid -> NaN
    ix = as_u32(x * 0x1p23f);                               // subnormal: normalise
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15;
  const int32_t k = static_cast<int32_t>(tmp) >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  double invc, logc;
  logf_tab(i, &invc, &logc);
  const double z = static_cast<double>(as_f32(iz));
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma(static_cast<double>(k), 0x1.62e42fefa39efp-1, logc);
  const double r2 = r * r;
  double y = __builtin_fma(r, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2);
  y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
  y = __builtin_fma(r2, y, y0 + r);
  return static_cast<float>(y);
}

// glibc 2.35 sysdeps/ieee754/flt-32/s_log1pf.c (fdlibm), as written
EM_FN float log1pf_general(float x) {
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, two25 = 3.355443200e+07f;
  const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f,
              Lp4 = 2.2222198546e-01f, Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f,
              Lp7 = 1.4798198640e-01f;
  const float zero = 0.0f;
  float hfsq, f = 0.0f, c = 0.0f, s, z, R, u;
  int32_t k, hx, hu = 0, ax;
  hx = static_cast<int32_t>(as_u32(x));
  ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3ed413d7) {  // x < 0.41422
    if (ax >= 0x3f800000) {  // x <= -1.0
      if (x == -1.0f) return -two25 / zero;
      return (x - x) / (x - x);
    }
    if (ax < 0x31000000) {  // |x| < 2**-29
      if (ax < 0x24800000) return x;  // |x| < 2**-54
      return x - x * x * 0.5f;
    }
    if (hx > 0 || hx <= static_cast<int32_t>(0xbe95f61f)) {
      k = 0;
      f = x;
      hu = 1;
    }  // -0.2929 < x < 0.41422
  }
  if (hx >= 0x7f800000) return x + x;
  if (k != 0) {
    if (hx < 0x5a000000) {
      u = 1.0f + x;
      hu = static_cast<int32_t>(as_u32(u));
      k = (hu >> 23) - 127;
      c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);
      c /= u;
    } else {
      u = x;
      hu = static_cast<int32_t>(as_u32(u));
      k = (hu >> 23) - 127;
      c = 0;
    }
    hu &= 0x007fffff;
    if (hu < 0x3504f7) {
      u = as_f32(static_cast<uint32_t>(hu | 0x3f800000));
    } else {
      k += 1;
      u = as_f32(static_cast<uint32_t>(hu | 0x3f000000));
      hu = (0x00800000 - hu) >> 2;
    }
    f = u - 1.0f;
  }
  hfsq = 0.5f * f * f;
  if (hu == 0) {  // |f| < 2**-20
    if (f == zero) {
      if (k == 0) return zero;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    R = hfsq * (1.0f - 0.66666666666666666f * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  s = f / (2.0f + f);
  z = s * s;
  R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// The same function with the common range in select form (see expm1f below): x finite, -1 < x < 2^53,
// |x| >= 2^-29, and a reduced argument that is not within 2^-20 of a power of two; everything else
// goes to log1pf_general.  The source's two argument classes -- x in (-0.2929, 0.41422), where f = x and
// k = 0, and the rest, where 1 + x is normalised to [sqrt(2)/2, sqrt(2)) -- are selects too: a wavefront's
// lanes are usually spread over both (a decoder's messages are), and a divergent branch then runs both
// sides one after the other plus the mask bookkeeping; computing the second class's reduction for every
// lane is cheaper (measured on the Tanh rule: tanhf_c9 below, same idea).
EM_FN float log1pf(float x) {
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
  const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f,
              Lp4 = 2.2222198546e-01f, Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f,
              Lp7 = 1.4798198640e-01f;
  const int32_t hx = static_cast<int32_t>(as_u32(x));
  const int32_t ax = hx & 0x7fffffff;
  // special: x <= -1 or NaN/negative huge (hx < 0 with ax >= 1.0), tiny, x >= 2^53 or inf/NaN
  if ((hx < 0 && ax >= 0x3f800000) || ax < 0x31000000 || hx >= 0x5a000000) return log1pf_general(x);
  const bool direct = hx < 0x3ed413d7 && (hx > 0 || hx <= static_cast<int32_t>(0xbe95f61f));  // -0.2929 < x < 0.41422
  // the second class's reduction, evaluated for every lane.  (Skipping it with a scalar branch when no
  // lane of the wavefront needs it was tried and lost: Minstarapproxf32 -10 %, the others unchanged.)
  const float u0 = 1.0f + x;
  int32_t hu = static_cast<int32_t>(as_u32(u0));
  int32_t k = (hu >> 23) - 127;
  float c = (k > 0) ? 1.0f - (u0 - x) : x - (u0 - 1.0f);
  c = fdiv_v<EM_FDIV_L1P_C>(c, u0);
  hu &= 0x007fffff;
  const bool low = hu < 0x3504f7;
  k += low ? 0 : 1;
  const float u = as_f32(static_cast<uint32_t>(hu | (low ? 0x3f800000 : 0x3f000000)));
  hu = low ? hu : ((0x00800000 - hu) >> 2);
  if (!direct && hu == 0) return log1pf_general(x);  // |f| < 2^-20 (rare)
  const float f = direct ? x : u - 1.0f;
  k = direct ? 0 : k;
  c = direct ? 0.0f : c;
  const float hfsq = 0.5f * f * f;
  const float s = fdiv_v<EM_FDIV_L1P_S>(f, 2.0f + f);
  const float z = s * s;
  const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  const float kf = static_cast<float>(k);
  const float r0 = f - (hfsq - s * (hfsq + R));
  const float rk = kf * ln2_hi - ((hfsq - (s * (hfsq + R) + (kf * ln2_lo + c))) - f);
  return (k == 0) ? r0 : rk;
}

// ln_1p(exp(-a)) for a >= 0: the correction term of the min* rules (arithmetic.rs:510, 965-966, where a is
// |x - y| or x + y of two magnitudes).  expf and log1pf above, fused for this argument range: exp(-a) is in
// (0, 1], so log1pf's classes reduce to "direct" (y < 0.41422) and "1 + y in [1.41422, 2]" (exponent 0, so the
// source's c is y - (u - 1) and k ends as 0 or 1), its tiny class returns y itself (y - y*y/2 rounds to y below
// 2^-29), and the k-dependent products become constants.  Per lane the operations are the two functions'; the
// rare arguments (a >= 87, NaN, 1 + y within 2^-20 of 2) go through them as written.  Checked against glibc's
// log1pf(expf(-a)) on every non-negative float.
EM_FN float corrf(float a) {
  if (!(a < 87.0f)) return log1pf(expf(-a));  // exp's underflow classes, infinity, NaN
  // expf(-a), main path
  const double xd = -static_cast<double>(a);
  const double invln2n = 0x1.71547652b82fep+5, shift = 0x1.8p+52;
  double kd = __builtin_fma(invln2n, xd, shift);
  const uint64_t ki = as_u64(kd);
  kd -= shift;
  const double r = __builtin_fma(invln2n, xd, -kd);
  const double sc = as_f64(exp2f_tab(static_cast<uint32_t>(ki & 31)) + (ki << 47));
  const double zz = __builtin_fma(r, 0x1.c6af84b912394p-20, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double yd = __builtin_fma(r, 0x1.62e42ff0c52d6p-6, 1.0);
  yd = __builtin_fma(zz, r2, yd);
  const float y = static_cast<float>(yd * sc);
  // log1pf(y), 0 < y <= 1
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
  const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f,
              Lp4 = 2.2222198546e-01f, Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f,
              Lp7 = 1.4798198640e-01f;
  const uint32_t hy = as_u32(y);
  const bool direct = hy < 0x3ed413d7u;
  const float u0 = 1.0f + y;
  const uint32_t hb = as_u32(u0);
  const uint32_t hu = hb & 0x007fffffu;
  const bool low = hu < 0x3504f7u;
  const uint32_t hu2 = low ? hu : ((0x00800000u - hu) >> 2);
  // 1 + y = 2 exactly (a = 0) or within 2^-20 of a power of two: the source's |f| < 2^-20 class
  if (!direct && ((hb >> 23) != 127u || hu2 == 0u)) return log1pf(y);
  float c = y - (u0 - 1.0f);
  c = fdiv_v<EM_FDIV_L1P_C>(c, u0);
  const float u = as_f32(hu | (low ? 0x3f800000u : 0x3f000000u));
  const float f = direct ? y : u - 1.0f;
  const bool kzero = direct || low;
  const float hfsq = 0.5f * f * f;
  const float s = fdiv_v<EM_FDIV_L1P_S>(f, 2.0f + f);
  const float z = s * s;
  const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  const float r0 = f - (hfsq - s * (hfsq + R));
  const float rk = ln2_hi - ((hfsq - (s * (hfsq + R) + (ln2_lo + c))) - f);
  const float res = kzero ? r0 : rk;
  return (hy < 0x31000000u) ? y : res;  // y < 2^-29: the source returns y - y*y/2, which is y
}

// glibc 2.35 sysdeps/ieee754/flt-32/s_expm1f.c (fdlibm), as written
EM_FN float expm1f_general(float x) {
  const float one = 1.0f, huge = 1.0e+30f, tiny = 1.0e-30f;
  const float o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
              invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
              Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  float y, hi, lo, c = 0.0f, t, e, hxs, hfx, r1;
  int32_t k, xsb;
  uint32_t hx = as_u32(x);
  xsb = static_cast<int32_t>(hx & 0x80000000u);
  hx &= 0x7fffffffu;
  if (hx >= 0x4195b844u) {      // |x| >= 27 ln2
    if (hx >= 0x42b17218u) {    // |x| >= 88.721...
      if (hx > 0x7f800000u) return x + x;
      if (hx == 0x7f800000u) return (xsb == 0) ? x : -1.0f;
      if (x > o_threshold) return huge * huge;
    }
    if (xsb != 0) return tiny - one;  // x < -27 ln2
  }
  if (hx > 0x3eb17218u) {    // |x| > 0.5 ln2
    if (hx < 0x3F851592u) {  // |x| < 1.5 ln2
      if (xsb == 0) {
        hi = x - ln2_hi;
        lo = ln2_lo;
        k = 1;
      } else {
        hi = x + ln2_hi;
        lo = -ln2_lo;
        k = -1;
      }
    } else {
      k = static_cast<int32_t>(invln2 * x + ((xsb == 0) ? 0.5f : -0.5f));
      t = static_cast<float>(k);
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x33000000u) {  // |x| < 2**-25
    t = huge + x;
    return x - (t - (huge + x));
  } else {
    k = 0;
  }
  hfx = 0.5f * x;
  hxs = x * hfx;
  r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  t = 3.0f - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0f - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5f * (x - e) - 0.5f;
  if (k == 1) {
    if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
    return one + 2.0f * (x - e);
  }
  if (k <= -2 || k > 56) {
    y = one - (e - x);
    y = as_f32(as_u32(y) + (static_cast<uint32_t>(k) << 23));
    return y - one;
  }
  if (k < 23) {
    t = as_f32(0x3f800000u - (0x1000000u >> k));  // 1 - 2^-k
    y = t - (e - x);
    y = as_f32(as_u32(y) + (static_cast<uint32_t>(k) << 23));
  } else {
    t = as_f32(static_cast<uint32_t>(0x7f - k) << 23);  // 2^-k
    y = x - (e + t);
    y += one;
    y = as_f32(as_u32(y) + (static_cast<uint32_t>(k) << 23));
  }
  return y;
}

// The same function with the common range (2^-25 <= |x| < 27 ln 2) in select form: a wavefront whose
// lanes fall into different k classes (reduction by 0, +-1, n ln2; six reconstruction formulas) ran
// every branch of the source one after the other, each with its own mask bookkeeping.  Here the
// reduction is one formula (k = 0 and k = +-1 are the general formula with t = 0 and t = +-1: the
// products t*ln2_hi / t*ln2_lo are exact) and the reconstructions are selected from straight-line
// candidates.  Per lane the selected value is the one the source computes; checked against glibc on
// all 2^32 arguments.  Everything else (NaN, infinities, overflow, tiny) goes to expm1f_general.
EM_FN float expm1f(float x) {
  const float one = 1.0f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
              Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  const uint32_t bits = as_u32(x);
  const uint32_t hx = bits & 0x7fffffffu;
  if (hx >= 0x4195b844u || hx < 0x33000000u) return expm1f_general(x);  // rare: not taken by a whole wave
  const bool neg = (bits >> 31) != 0;
  // argument reduction (s_expm1f.c: "if |x| > 0.5 ln2 ... else k = 0")
  float kf = 0.0f;
  if (hx > 0x3eb17218u) {
    const float general = static_cast<float>(static_cast<int32_t>(invln2 * x + (neg ? -0.5f : 0.5f)));
    kf = (hx < 0x3F851592u) ? (neg ? -1.0f : 1.0f) : general;
  }
  const int32_t k = static_cast<int32_t>(kf);
  const float hi = x - kf * ln2_hi;
  const float lo = kf * ln2_lo;
  const float xr = hi - lo;
  const float c = (hi - xr) - lo;
  // primary range
  const float hfx = 0.5f * xr;
  const float hxs = xr * hfx;
  const float r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  const float t = 3.0f - r1 * hfx;
  const float e = hxs * fdiv_v<EM_FDIV_EXPM1>(r1 - t, 6.0f - xr * t);
  // reconstruction candidates
  const float r0 = xr - (xr * e - hxs);                                  // k == 0
  const float e2 = (xr * (e - c) - c) - hxs;
  const float rm1 = 0.5f * (xr - e2) - 0.5f;                             // k == -1
  const float rp1 = (xr < -0.25f) ? -2.0f * (e2 - (xr + 0.5f)) : one + 2.0f * (xr - e2);  // k == 1
  const float dd = e2 - xr;
  const uint32_t kbits = static_cast<uint32_t>(k) << 23;
  const float ya = as_f32(as_u32(one - dd) + kbits) - one;               // k <= -2 (k > 56 is outside this range)
  const uint32_t ksmall = (k >= 2 && k < 23) ? static_cast<uint32_t>(k) : 2u;
  const float t1 = as_f32(0x3f800000u - (0x1000000u >> ksmall));         // 1 - 2^-k
  const float yb = as_f32(as_u32(t1 - dd) + kbits);                      // 2 <= k < 23
  const uint32_t klarge = (k >= 23 && k <= 127) ? static_cast<uint32_t>(k) : 23u;
  const float t2 = as_f32((0x7fu - klarge) << 23);                       // 2^-k
  const float yc = as_f32(as_u32((xr - (e2 + t2)) + one) + kbits);       // k >= 23
  float r = (k < 23) ? yb : yc;
  r = (k <= -2) ? ya : r;
  r = (k == 1) ? rp1 : r;
  r = (k == -1) ? rm1 : r;
  r = (k == 0) ? r0 : r;
  return r;
}

// glibc 2.35 sysdeps/ieee754/flt-32/s_tanhf.c (fdlibm).  The two branches of the source
// (|x| >= 1: 1 - 2/(expm1(2|x|) + 2); |x| < 1: -t/(t + 2), t = expm1(-2|x|)) share one expm1f
// evaluation and one division, and the whole function is one straight line of selects: a wavefront whose
// lanes fall into different classes (they do: the arguments are a decoder's messages) runs no divergent
// branch.  expm1f is inlined in the form its two callers' arguments need: 2|x| >= 2 gives k >= 3,
// -2|x| in (-2, 0] gives k in -3..0, so the k = 1 reconstruction never occurs; |x| >= 22 (infinity
// included) selects the source's 1 - tiny = 1 at the end, computed on a clamped argument; NaN is passed
// through; x = +-0 and |x| < 2^-55 need no case of their own (expm1f's tiny class returns its argument and
// 2a / (2 - 2a) = a there).  Per lane the operations are the source's; checked against glibc on all 2^32
// arguments on the device.
EM_FN float tanhf(float x) {
  const float one = 1.0f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
              Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  const uint32_t jx = as_u32(x);
  const uint32_t ix = jx & 0x7fffffffu;
  const bool sat = ix >= 0x41b00000u;                 // |x| >= 22, infinity, NaN
  const float ax = as_f32(sat ? 0x41a00000u : ix);    // the common path runs on 20.0 there (result discarded)
  const bool big = ix >= 0x3f800000u;                 // |x| >= 1
  const float arg = ax * (big ? 2.0f : -2.0f);        // expm1f's argument (exact)
  const uint32_t hx = as_u32(arg) & 0x7fffffffu;
  const float g = static_cast<float>(static_cast<int32_t>(invln2 * arg + (big ? 0.5f : -0.5f)));
  float kf = (hx < 0x3F851592u) ? -1.0f : g;
  kf = (hx > 0x3eb17218u) ? kf : 0.0f;
  const int32_t k = static_cast<int32_t>(kf);
  const float hi = arg - kf * ln2_hi;
  const float lo = kf * ln2_lo;
  const float xr = hi - lo;
  const float c = (hi - xr) - lo;
  const float hfx = 0.5f * xr;
  const float hxs = xr * hfx;
  const float r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  const float t3 = 3.0f - r1 * hfx;
  const float e = hxs * fdiv_v<EM_FDIV_EXPM1>(r1 - t3, 6.0f - xr * t3);
  const float r0 = xr - (xr * e - hxs);                                   // k == 0
  const float e2 = (xr * (e - c) - c) - hxs;
  const float rm1 = 0.5f * (xr - e2) - 0.5f;                              // k == -1
  const float dd = e2 - xr;
  const uint32_t kbits = static_cast<uint32_t>(k) << 23;
  const float ya = as_f32(as_u32(one - dd) + kbits) - one;                // k <= -2 or k > 56
  const uint32_t ksmall = (k >= 2 && k < 23) ? static_cast<uint32_t>(k) : 2u;
  const float t1 = as_f32(0x3f800000u - (0x1000000u >> ksmall));          // 1 - 2^-k
  const float yb = as_f32(as_u32(t1 - dd) + kbits);                       // 3 <= k < 23
  const uint32_t klarge = (k >= 23 && k <= 56) ? static_cast<uint32_t>(k) : 23u;
  const float t2 = as_f32((0x7fu - klarge) << 23);                        // 2^-k
  const float yc = as_f32(as_u32((xr - (e2 + t2)) + one) + kbits);        // 23 <= k <= 56
  float r_big = (k < 23) ? yb : yc;
  r_big = (k > 56) ? ya : r_big;
  const float r_small = (k == 0) ? r0 : ((k == -1) ? rm1 : ya);
  float t = big ? r_big : r_small;
  t = (hx < 0x33000000u) ? arg : t;                                       // |arg| < 2^-25: expm1f returns its argument
  const float q = fdiv_v<EM_FDIV_TANH>(big ? 2.0f : -t, t + 2.0f);
  float z = big ? one - q : q;                                            // z >= +0
  z = sat ? one : z;                                                      // one - tiny
  const float r = as_f32(as_u32(z) | (jx & 0x80000000u));
  return (ix > 0x7f800000u) ? x + x : r;                                  // NaN
}

// The Phi rule's function, -ln(tanh(max(x, 1e-30) / 2)) (arithmetic.rs:180-186): tanhf and logf above, fused for
// this argument.  The argument of tanhf is h >= 5e-31 (or +inf; a NaN x becomes 1e-30 in the max), so its sign
// and NaN handling drop out; its result is a normal number in (0, 1], so logf's classes (zero, negative,
// infinity, NaN, subnormal) cannot occur and only the source's "x == 1 -> +0" remains, as a select.  Per lane the
// operations are the two functions'.  Checked against glibc's -logf(tanhf(.)) on all 2^32 arguments.
EM_FN float phif(float x) {
  const float one = 1.0f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
              Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  // (fmaxf, not the builtin: a signalling NaN then becomes a NaN on the host as on the device, and is passed on)
  const float h = 0.5f * fmaxf(x, 1e-30f);
  // tanhf(h), h > 0
  const uint32_t ix = as_u32(h);
  const bool sat = ix >= 0x41b00000u;                 // h >= 22, infinity
  const float ax = as_f32(sat ? 0x41a00000u : ix);
  const bool big = ix >= 0x3f800000u;                 // h >= 1
  const float arg = ax * (big ? 2.0f : -2.0f);
  const uint32_t hx = as_u32(arg) & 0x7fffffffu;
  const float g = static_cast<float>(static_cast<int32_t>(invln2 * arg + (big ? 0.5f : -0.5f)));
  float kf = (hx < 0x3F851592u) ? -1.0f : g;
  kf = (hx > 0x3eb17218u) ? kf : 0.0f;
  const int32_t k = static_cast<int32_t>(kf);
  const float hi = arg - kf * ln2_hi;
  const float lo = kf * ln2_lo;
  const float xr = hi - lo;
  const float c = (hi - xr) - lo;
  const float hfx = 0.5f * xr;
  const float hxs = xr * hfx;
  const float r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  const float t3 = 3.0f - r1 * hfx;
  const float e = hxs * fdiv_v<EM_FDIV_EXPM1>(r1 - t3, 6.0f - xr * t3);
  const float r0 = xr - (xr * e - hxs);
  const float e2 = (xr * (e - c) - c) - hxs;
  const float rm1 = 0.5f * (xr - e2) - 0.5f;
  const float dd = e2 - xr;
  const uint32_t kbits = static_cast<uint32_t>(k) << 23;
  const float ya = as_f32(as_u32(one - dd) + kbits) - one;
  const uint32_t ksmall = (k >= 2 && k < 23) ? static_cast<uint32_t>(k) : 2u;
  const float t1 = as_f32(0x3f800000u - (0x1000000u >> ksmall));
  const float yb = as_f32(as_u32(t1 - dd) + kbits);
  const uint32_t klarge = (k >= 23 && k <= 56) ? static_cast<uint32_t>(k) : 23u;
  const float t2 = as_f32((0x7fu - klarge) << 23);
  const float yc = as_f32(as_u32((xr - (e2 + t2)) + one) + kbits);
  float r_big = (k < 23) ? yb : yc;
  r_big = (k > 56) ? ya : r_big;
  const float r_small = (k == 0) ? r0 : ((k == -1) ? rm1 : ya);
  float t = big ? r_big : r_small;
  t = (hx < 0x33000000u) ? arg : t;
  const float q = fdiv_v<EM_FDIV_TANH>(big ? 2.0f : -t, t + 2.0f);
  float th = big ? one - q : q;
  th = sat ? one : th;
  // logf(th), th normal in (0, 1]
  const uint32_t iy = as_u32(th);
  const uint32_t tmp = iy - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15;
  const int32_t kl = static_cast<int32_t>(tmp) >> 23;
  const uint32_t iz = iy - (tmp & 0xff800000u);
  double invc, logc;
  logf_tab(i, &invc, &logc);
  const double z = static_cast<double>(as_f32(iz));
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma(static_cast<double>(kl), 0x1.62e42fefa39efp-1, logc);
  const double r2 = r * r;
  double y = __builtin_fma(r, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2);
  y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
  y = __builtin_fma(r2, y, y0 + r);
  const float lg = (iy == 0x3f800000u) ? 0.0f : static_cast<float>(y);
  return (h != h) ? h : -lg;  // NaN only for a signalling-NaN x (the max turns a quiet one into 1e-30)
}

// tanhf restricted to finite |x| <= 9 -- what the Tanh rule hands it after its clamp
// (arithmetic.rs:357, 435).  For these arguments every special case of tanhf / expm1f either cannot occur
// (NaN, infinity, |x| >= 22, expm1f's k = 1 and overflow classes: the argument of expm1f is 2|x| >= 2 or
// -2|x| in (-2, 0], so k is 3..26 or -3..0) or is reproduced by the common path itself (x = +-0 and
// |x| < 2^-55: expm1f's tiny class returns its argument, and 2a / (2 - 2a) = a there), so the function is one
// straight line of selects: no divergent branches in a wavefront whose lanes are in different classes.
// Per lane the operations are those of tanhf above; checked against glibc on every float in [-9, 9].
EM_FN float tanhf_c9(float x) {
  const float one = 1.0f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
              Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  const uint32_t jx = as_u32(x);
  const uint32_t ix = jx & 0x7fffffffu;
  const float ax = as_f32(ix);
  const bool big = ix >= 0x3f800000u;                 // |x| >= 1
  const float arg = ax * (big ? 2.0f : -2.0f);        // expm1f's argument (exact)
  const uint32_t hx = as_u32(arg) & 0x7fffffffu;
  // s_expm1f.c argument reduction: k = 0 up to 0.5 ln2, -1 up to 1.5 ln2 (only negative arguments are that
  // small here), else trunc(invln2 * x +- 0.5)
  const float g = static_cast<float>(static_cast<int32_t>(invln2 * arg + (big ? 0.5f : -0.5f)));
  float kf = (hx < 0x3F851592u) ? -1.0f : g;
  kf = (hx > 0x3eb17218u) ? kf : 0.0f;
  const int32_t k = static_cast<int32_t>(kf);
  const float hi = arg - kf * ln2_hi;
  const float lo = kf * ln2_lo;
  const float xr = hi - lo;
  const float c = (hi - xr) - lo;
  const float hfx = 0.5f * xr;
  const float hxs = xr * hfx;
  const float r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  const float t3 = 3.0f - r1 * hfx;
  const float e = hxs * fdiv_v<EM_FDIV_EXPM1>(r1 - t3, 6.0f - xr * t3);
  const float r0 = xr - (xr * e - hxs);                                   // k == 0
  const float e2 = (xr * (e - c) - c) - hxs;
  const float rm1 = 0.5f * (xr - e2) - 0.5f;                              // k == -1
  const float dd = e2 - xr;
  const uint32_t kbits = static_cast<uint32_t>(k) << 23;
  const float ya = as_f32(as_u32(one - dd) + kbits) - one;                // k <= -2
  const uint32_t ksmall = (k >= 2 && k < 23) ? static_cast<uint32_t>(k) : 2u;
  const float t1 = as_f32(0x3f800000u - (0x1000000u >> ksmall));          // 1 - 2^-k
  const float yb = as_f32(as_u32(t1 - dd) + kbits);                       // 3 <= k < 23
  const uint32_t klarge = (k >= 23) ? static_cast<uint32_t>(k) : 23u;
  const float t2 = as_f32((0x7fu - klarge) << 23);                        // 2^-k
  const float yc = as_f32(as_u32((xr - (e2 + t2)) + one) + kbits);        // k >= 23
  const float r_big = (k < 23) ? yb : yc;
  const float r_small = (k == 0) ? r0 : ((k == -1) ? rm1 : ya);
  float t = big ? r_big : r_small;
  t = (hx < 0x33000000u) ? arg : t;                                       // |arg| < 2^-25: expm1f returns its argument
  const float q = fdiv_v<EM_FDIV_TANH>(big ? 2.0f : -t, t + 2.0f);
  const float z = big ? one - q : q;                                      // z >= +0
  return as_f32(as_u32(z) | (jx & 0x80000000u));
}

// Rust std's f32::atanh, 0.5 * ln_1p(2x / (1 - x)) (library/std/src/f32.rs), as the Tanh rule calls it
// (arithmetic.rs:376) -- a function of one argument, so it is checked exhaustively like the others
EM_FN float atanh_rs(float x) {
  // +-1, beyond, NaN: IEEE division
  if (!(__builtin_fabsf(x) < 1.0f)) return 0.5f * log1pf((2.0f * x) / (1.0f - x));
  // +-0: the quotient is the argument itself (the short sequence would lose the sign of a zero quotient)
  const float q = fdiv_v<EM_FDIV_ATANH>(2.0f * x, 1.0f - x);
  return 0.5f * log1pf((x == 0.0f) ? x : q);
}

// atanh_rs without its rare classes, for callers that unroll it many times (the slice-persistent layered kernel): one
// straight line of selects.  *rare is set for the arguments this line does not cover -- NaN or |x| >= 1, a quotient
// 2x / (1 - x) at or below -1 or beyond log1pf's huge class, and log1pf's "reduced argument within 2^-20 of a power of
// two" class -- and the caller then takes atanh_rs(x) instead (the value returned here is unspecified).  log1pf's tiny
// class (|argument| < 2^-29, which includes the zero quotient of a zero product) is folded in as two selects, so that
// idle lanes computing on zeros do not send their wavefront down the rare path.  Per lane the operations are those of
// atanh_rs; checked on all 2^32 arguments (tools/check_exact_math_device.hip, "atanh main").
EM_FN float atanh_rs_main(float x, bool *rare) {
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
  const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f,
              Lp4 = 2.2222198546e-01f, Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f,
              Lp7 = 1.4798198640e-01f;
  const float q = fdiv_v<EM_FDIV_ATANH>(2.0f * x, 1.0f - x);
  const float a = (x == 0.0f) ? x : q;  // log1pf's argument
  const int32_t hx = static_cast<int32_t>(as_u32(a));
  const int32_t ax = hx & 0x7fffffff;
  const bool tiny = ax < 0x31000000;  // |a| < 2^-29
  bool odd = !(__builtin_fabsf(x) < 1.0f) || (hx < 0 && ax >= 0x3f800000) || hx >= 0x5a000000;
  const bool direct = hx < 0x3ed413d7 && (hx > 0 || hx <= static_cast<int32_t>(0xbe95f61f));  // -0.2929 < a < 0.41422
  const float u0 = 1.0f + a;
  int32_t hu = static_cast<int32_t>(as_u32(u0));
  int32_t k = (hu >> 23) - 127;
  float c = (k > 0) ? 1.0f - (u0 - a) : a - (u0 - 1.0f);
  c = fdiv_v<EM_FDIV_L1P_C>(c, u0);
  hu &= 0x007fffff;
  const bool low = hu < 0x3504f7;
  k += low ? 0 : 1;
  const float u = as_f32(static_cast<uint32_t>(hu | (low ? 0x3f800000 : 0x3f000000)));
  hu = low ? hu : ((0x00800000 - hu) >> 2);
  odd = odd || (!tiny && !direct && hu == 0);  // |f| < 2^-20
  const float f = direct ? a : u - 1.0f;
  k = direct ? 0 : k;
  c = direct ? 0.0f : c;
  const float hfsq = 0.5f * f * f;
  const float s = fdiv_v<EM_FDIV_L1P_S>(f, 2.0f + f);
  const float z = s * s;
  const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  const float kf = static_cast<float>(k);
  const float r0 = f - (hfsq - s * (hfsq + R));
  const float rk = kf * ln2_hi - ((hfsq - (s * (hfsq + R) + (kf * ln2_lo + c))) - f);
  const float main_path = (k == 0) ? r0 : rk;
  // s_log1pf.c: |x| < 2^-54 returns x, |x| < 2^-29 returns x - x*x*0.5
  const float small = (ax < 0x24800000) ? a : a - a * a * 0.5f;
  *rare = odd;
  return 0.5f * (tiny ? small : main_path);
}

// ---------------------------------------------------------------------------------------------
// Double precision: exp, log (Szabolcs Nagy's routines, x86-64 FMA build of glibc 2.35:
// sysdeps/ieee754/dbl-64/e_exp.c, e_log.c) and log1p, expm1, tanh (fdlibm: s_log1p.c, s_expm1.c,
// s_tanh.c).  Checked against the host libm on billions of arguments (tools/check_exact_math.cpp).
// ---------------------------------------------------------------------------------------------

EM_FN uint64_t exp_tab(uint32_t i) {
  constexpr uint64_t T[256] = {
0x0000000000000000ULL, 0x3ff0000000000000ULL, 0x3c9b3b4f1a88bf6eULL, 0x3feff63da9fb3335ULL,
0xbc7160139cd8dc5dULL, 0x3fefec9a3e778061ULL, 0xbc905e7a108766d1ULL, 0x3fefe315e86e7f85ULL,
0x3c8cd2523567f613ULL, 0x3fefd9b0d3158574ULL, 0xbc8bce8023f98efaULL, 0x3fefd06b29ddf6deULL,
0x3c60f74e61e6c861ULL, 0x3fefc74518759bc8ULL, 0x3c90a3e45b33d399ULL, 0x3fefbe3ecac6f383ULL,
0x3c979aa65d837b6dULL, 0x3fefb5586cf9890fULL, 0x3c8eb51a92fdeffcULL, 0x3fefac922b7247f7ULL,
0x3c3ebe3d702f9cd1ULL, 0x3fefa3ec32d3d1a2ULL, 0xbc6a033489906e0bULL, 0x3fef9b66affed31bULL,
0xbc9556522a2fbd0eULL, 0x3fef9301d0125b51ULL, 0xbc5080ef8c4eea55ULL, 0x3fef8abdc06c31ccULL,
0xbc91c923b9d5f416ULL, 0x3fef829aaea92de0ULL, 0x3c80d3e3e95c55afULL, 0x3fef7a98c8a58e51ULL,
0xbc801b15eaa59348ULL, 0x3fef72b83c7d517bULL, 0xbc8f1ff055de323dULL, 0x3fef6af9388c8deaULL,
0x3c8b898c3f1353bfULL, 0x3fef635beb6fcb75ULL, 0xbc96d99c7611eb26ULL, 0x3fef5be084045cd4ULL,
0x3c9aecf73e3a2f60ULL, 0x3fef54873168b9aaULL, 0xbc8fe782cb86389dULL, 0x3fef4d5022fcd91dULL,
0x3c8a6f4144a6c38dULL, 0x3fef463b88628cd6ULL, 0x3c807a05b0e4047dULL, 0x3fef3f49917ddc96ULL,
0x3c968efde3a8a894ULL, 0x3fef387a6e756238ULL, 0x3c875e18f274487dULL, 0x3fef31ce4fb2a63fULL,
0x3c80472b981fe7f2ULL, 0x3fef2b4565e27cddULL, 0xbc96b87b3f71085eULL, 0x3fef24dfe1f56381ULL,
0x3c82f7e16d09ab31ULL, 0x3fef1e9df51fdee1ULL, 0xbc3d219b1a6fbffaULL, 0x3fef187fd0dad990ULL,
0x3c8b3782720c0ab4ULL, 0x3fef1285a6e4030bULL, 0x3c6e149289cecb8fULL, 0x3fef0cafa93e2f56ULL,
0x3c834d754db0abb6ULL, 0x3fef06fe0a31b715ULL, 0x3c864201e2ac744cULL, 0x3fef0170fc4cd831ULL,
0x3c8fdd395dd3f84aULL, 0x3feefc08b26416ffULL, 0xbc86a3803b8e5b04ULL, 0x3feef6c55f929ff1ULL,
0xbc924aedcc4b5068ULL, 0x3feef1a7373aa9cbULL, 0xbc9907f81b512d8eULL, 0x3feeecae6d05d866ULL,
0xbc71d1e83e9436d2ULL, 0x3feee7db34e59ff7ULL, 0xbc991919b3ce1b15ULL, 0x3feee32dc313a8e5ULL,
0x3c859f48a72a4c6dULL, 0x3feedea64c123422ULL, 0xbc9312607a28698aULL, 0x3feeda4504ac801cULL,
0xbc58a78f4817895bULL, 0x3feed60a21f72e2aULL, 0xbc7c2c9b67499a1bULL, 0x3feed1f5d950a897ULL,
0x3c4363ed60c2ac11ULL, 0x3feece086061892dULL, 0x3c9666093b0664efULL, 0x3feeca41ed1d0057ULL,
0x3c6ecce1daa10379ULL, 0x3feec6a2b5c13cd0ULL, 0x3c93ff8e3f0f1230ULL, 0x3feec32af0d7d3deULL,
0x3c7690cebb7aafb0ULL, 0x3feebfdad5362a27ULL, 0x3c931dbdeb54e077ULL, 0x3feebcb299fddd0dULL,
0xbc8f94340071a38eULL, 0x3feeb9b2769d2ca7ULL, 0xbc87deccdc93a349ULL, 0x3feeb6daa2cf6642ULL,
0xbc78dec6bd0f385fULL, 0x3feeb42b569d4f82ULL, 0xbc861246ec7b5cf6ULL, 0x3feeb1a4ca5d920fULL,
0x3c93350518fdd78eULL, 0x3feeaf4736b527daULL, 0x3c7b98b72f8a9b05ULL, 0x3feead12d497c7fdULL,
0x3c9063e1e21c5409ULL, 0x3feeab07dd485429ULL, 0x3c34c7855019c6eaULL, 0x3feea9268a5946b7ULL,
0x3c9432e62b64c035ULL, 0x3feea76f15ad2148ULL, 0xbc8ce44a6199769fULL, 0x3feea5e1b976dc09ULL,
0xbc8c33c53bef4da8ULL, 0x3feea47eb03a5585ULL, 0xbc845378892be9aeULL, 0x3feea34634ccc320ULL,
0xbc93cedd78565858ULL, 0x3feea23882552225ULL, 0x3c5710aa807e1964ULL, 0x3feea155d44ca973ULL,
0xbc93b3efbf5e2228ULL, 0x3feea09e667f3bcdULL, 0xbc6a12ad8734b982ULL, 0x3feea012750bdabfULL,
0xbc6367efb86da9eeULL, 0x3fee9fb23c651a2fULL, 0xbc80dc3d54e08851ULL, 0x3fee9f7df9519484ULL,
0xbc781f647e5a3ecfULL, 0x3fee9f75e8ec5f74ULL, 0xbc86ee4ac08b7db0ULL, 0x3fee9f9a48a58174ULL,
0xbc8619321e55e68aULL, 0x3fee9feb564267c9ULL, 0x3c909ccb5e09d4d3ULL, 0x3feea0694fde5d3fULL,
0xbc7b32dcb94da51dULL, 0x3feea11473eb0187ULL, 0x3c94ecfd5467c06bULL, 0x3feea1ed0130c132ULL,
0x3c65ebe1abd66c55ULL, 0x3feea2f336cf4e62ULL, 0xbc88a1c52fb3cf42ULL, 0x3feea427543e1a12ULL,
0xbc9369b6f13b3734ULL, 0x3feea589994cce13ULL, 0xbc805e843a19ff1eULL, 0x3feea71a4623c7adULL,
0xbc94d450d872576eULL, 0x3feea8d99b4492edULL, 0x3c90ad675b0e8a00ULL, 0x3feeaac7d98a6699ULL,
0x3c8db72fc1f0eab4ULL, 0x3feeace5422aa0dbULL, 0xbc65b6609cc5e7ffULL, 0x3feeaf3216b5448cULL,
0x3c7bf68359f35f44ULL, 0x3feeb1ae99157736ULL, 0xbc93091fa71e3d83ULL, 0x3feeb45b0b91ffc6ULL,
0xbc5da9b88b6c1e29ULL, 0x3feeb737b0cdc5e5ULL, 0xbc6c23f97c90b959ULL, 0x3feeba44cbc8520fULL,
0xbc92434322f4f9aaULL, 0x3feebd829fde4e50ULL, 0xbc85ca6cd7668e4bULL, 0x3feec0f170ca07baULL,
0x3c71affc2b91ce27ULL, 0x3feec49182a3f090ULL, 0x3c6dd235e10a73bbULL, 0x3feec86319e32323ULL,
0xbc87c50422622263ULL, 0x3feecc667b5de565ULL, 0x3c8b1c86e3e231d5ULL, 0x3feed09bec4a2d33ULL,
0xbc91bbd1d3bcbb15ULL, 0x3feed503b23e255dULL, 0x3c90cc319cee31d2ULL, 0x3feed99e1330b358ULL,
0x3c8469846e735ab3ULL, 0x3feede6b5579fdbfULL, 0xbc82dfcd978e9db4ULL, 0x3feee36bbfd3f37aULL,
0x3c8c1a7792cb3387ULL, 0x3feee89f995ad3adULL, 0xbc907b8f4ad1d9faULL, 0x3feeee07298db666ULL,
0xbc55c3d956dcaebaULL, 0x3feef3a2b84f15fbULL, 0xbc90a40e3da6f640ULL, 0x3feef9728de5593aULL,
0xbc68d6f438ad9334ULL, 0x3feeff76f2fb5e47ULL, 0xbc91eee26b588a35ULL, 0x3fef05b030a1064aULL,
0x3c74ffd70a5fddcdULL, 0x3fef0c1e904bc1d2ULL, 0xbc91bdfbfa9298acULL, 0x3fef12c25bd71e09ULL,
0x3c736eae30af0cb3ULL, 0x3fef199bdd85529cULL, 0x3c8ee3325c9ffd94ULL, 0x3fef20ab5fffd07aULL,
0x3c84e08fd10959acULL, 0x3fef27f12e57d14bULL, 0x3c63cdaf384e1a67ULL, 0x3fef2f6d9406e7b5ULL,
0x3c676b2c6c921968ULL, 0x3fef3720dcef9069ULL, 0xbc808a1883ccb5d2ULL, 0x3fef3f0b555dc3faULL,
0xbc8fad5d3ffffa6fULL, 0x3fef472d4a07897cULL, 0xbc900dae3875a949ULL, 0x3fef4f87080d89f2ULL,
0x3c74a385a63d07a7ULL, 0x3fef5818dcfba487ULL, 0xbc82919e2040220fULL, 0x3fef60e316c98398ULL,
0x3c8e5a50d5c192acULL, 0x3fef69e603db3285ULL, 0x3c843a59ac016b4bULL, 0x3fef7321f301b460ULL,
0xbc82d52107b43e1fULL, 0x3fef7c97337b9b5fULL, 0xbc892ab93b470dc9ULL, 0x3fef864614f5a129ULL,
0x3c74b604603a88d3ULL, 0x3fef902ee78b3ff6ULL, 0x3c83c5ec519d7271ULL, 0x3fef9a51fbc74c83ULL,
0xbc8ff7128fd391f0ULL, 0x3fefa4afa2a490daULL, 0xbc8dae98e223747dULL, 0x3fefaf482d8e67f1ULL,
0x3c8ec3bc41aa2008ULL, 0x3fefba1bee615a27ULL, 0x3c842b94c3a9eb32ULL, 0x3fefc52b376bba97ULL,
0x3c8a64a931d185eeULL, 0x3fefd0765b6e4540ULL, 0xbc8e37bae43be3edULL, 0x3fefdbfdad9cbe14ULL,
0x3c77893b4d91cd9dULL, 0x3fefe7c1819e90d8ULL, 0x3c5305c14160cc89ULL, 0x3feff3c22b8f71f1ULL};
  return T[i];
}

EM_FN double exp_special(double tmp, uint64_t sbits, uint64_t ki) {
  if ((ki & 0x80000000u) == 0) {
    // k > 0: the exponent of scale might have overflowed by <= 460
    sbits -= 1009ull << 52;
    const double scale = as_f64(sbits);
    return 0x1p1009 * __builtin_fma(scale, tmp, scale);
  }
  // k < 0: special care in the subnormal range
  sbits += 1022ull << 52;
  const double scale = as_f64(sbits);
  double y = scale + scale * tmp;
  if (y < 1.0) {
    double lo = scale - y + scale * tmp;
    const double hi = 1.0 + y;
    lo = 1.0 - hi + y + lo;
    y = (hi + lo) - 1.0;
    if (y == 0.0) y = 0.0;
  }
  return 0x1p-1022 * y;
}

EM_FN double exp(double x) {
  const uint64_t ix = as_u64(x);
  uint32_t abstop = static_cast<uint32_t>(ix >> 52) & 0x7ff;
  if (abstop - 0x3c9 >= 0x3f) {  // |x| < 2^-54 or |x| >= 512 or NaN
    if (abstop - 0x3c9 >= 0x80000000u) return 1.0 + x;  // tiny
    if (abstop >= 0x409) {                              // |x| >= 1024
      if (ix == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ff) return 1.0 + x;
      if (ix >> 63) return 0x1p-767 * 0x1p-767;          // underflow
      return 0x1p769 * 0x1p769;                          // overflow
    }
    abstop = 0;  // large x is special cased below
  }
  const double invln2n = 0x1.71547652b82fep+7, shift = 0x1.8p+52;
  double kd = __builtin_fma(x, invln2n, shift);
  const uint64_t ki = as_u64(kd);
  kd -= shift;
  double r = __builtin_fma(kd, -0x1.62e42fefa0000p-8, x);
  r = __builtin_fma(kd, -0x1.cf79abc9e3b3ap-47, r);
  const uint32_t idx = 2 * static_cast<uint32_t>(ki & 127);
  const uint64_t top = ki << 45;
  const double tail = as_f64(exp_tab(idx));
  const uint64_t sbits = exp_tab(idx + 1) + top;
  const double r2 = r * r;
  const double p23 = __builtin_fma(r, 0x1.555555555543cp-3, 0x1.ffffffffffdbdp-2);
  const double p45 = __builtin_fma(r, 0x1.1111167a4d017p-7, 0x1.55555cf172b91p-5);
  double tmp = __builtin_fma(p23, r2, r + tail);
  tmp = __builtin_fma(r2 * r2, p45, tmp);
  if (abstop == 0) return exp_special(tmp, sbits, ki);
  const double scale = as_f64(sbits);
  return __builtin_fma(scale, tmp, scale);
}

EM_FN void log_tab(uint32_t i, double *invc, double *logc) {
  constexpr uint64_t T[256] = {
0x3ff734f0c3e0de9fULL, 0xbfd7cc7f79e69000ULL, 0x3ff713786a2ce91fULL, 0xbfd76feec20d0000ULL,
0x3ff6f26008fab5a0ULL, 0xbfd713e31351e000ULL, 0x3ff6d1a61f138c7dULL, 0xbfd6b85b38287800ULL,
0x3ff6b1490bc5b4d1ULL, 0xbfd65d5590807800ULL, 0x3ff69147332f0cbaULL, 0xbfd602d076180000ULL,
0x3ff6719f18224223ULL, 0xbfd5a8ca86909000ULL, 0x3ff6524f99a51ed9ULL, 0xbfd54f4356035000ULL,
0x3ff63356aa8f24c4ULL, 0xbfd4f637c36b4000ULL, 0x3ff614b36b9ddc14ULL, 0xbfd49da7fda85000ULL,
0x3ff5f66452c65c4cULL, 0xbfd445923989a800ULL, 0x3ff5d867b5912c4fULL, 0xbfd3edf439b0b800ULL,
0x3ff5babccb5b90deULL, 0xbfd396ce448f7000ULL, 0x3ff59d61f2d91a78ULL, 0xbfd3401e17bda000ULL,
0x3ff5805612465687ULL, 0xbfd2e9e2ef468000ULL, 0x3ff56397cee76bd3ULL, 0xbfd2941b3830e000ULL,
0x3ff54725e2a77f93ULL, 0xbfd23ec58cda8800ULL, 0x3ff52aff42064583ULL, 0xbfd1e9e129279000ULL,
0x3ff50f22dbb2bddfULL, 0xbfd1956d2b48f800ULL, 0x3ff4f38f4734ded7ULL, 0xbfd141679ab9f800ULL,
0x3ff4d843cfde2840ULL, 0xbfd0edd094ef9800ULL, 0x3ff4bd3ec078a3c8ULL, 0xbfd09aa518db1000ULL,
0x3ff4a27fc3e0258aULL, 0xbfd047e65263b800ULL, 0x3ff4880524d48434ULL, 0xbfcfeb224586f000ULL,
0x3ff46dce1b192d0bULL, 0xbfcf474a7517b000ULL, 0x3ff453d9d3391854ULL, 0xbfcea4443d103000ULL,
0x3ff43a2744b4845aULL, 0xbfce020d44e9b000ULL, 0x3ff420b54115f8fbULL, 0xbfcd60a22977f000ULL,
0x3ff40782da3ef4b1ULL, 0xbfccc00104959000ULL, 0x3ff3ee8f5d57fe8fULL, 0xbfcc202956891000ULL,
0x3ff3d5d9a00b4ce9ULL, 0xbfcb81178d811000ULL, 0x3ff3bd60c010c12bULL, 0xbfcae2c9ccd3d000ULL,
0x3ff3a5242b75dab8ULL, 0xbfca45402e129000ULL, 0x3ff38d22cd9fd002ULL, 0xbfc9a877681df000ULL,
0x3ff3755bc5847a1cULL, 0xbfc90c6d69483000ULL, 0x3ff35dce49ad36e2ULL, 0xbfc87120a645c000ULL,
0x3ff34679984dd440ULL, 0xbfc7d68fb4143000ULL, 0x3ff32f5cceffcb24ULL, 0xbfc73cb83c627000ULL,
0x3ff3187775a10d49ULL, 0xbfc6a39a9b376000ULL, 0x3ff301c8373e3990ULL, 0xbfc60b3154b7a000ULL,
0x3ff2eb4ebb95f841ULL, 0xbfc5737d76243000ULL, 0x3ff2d50a0219a9d1ULL, 0xbfc4dc7b8fc23000ULL,
0x3ff2bef9a8b7fd2aULL, 0xbfc4462c51d20000ULL, 0x3ff2a91c7a0c1babULL, 0xbfc3b08abc830000ULL,
0x3ff293726014b530ULL, 0xbfc31b996b490000ULL, 0x3ff27dfa5757a1f5ULL, 0xbfc2875490a44000ULL,
0x3ff268b39b1d3bbfULL, 0xbfc1f3b9f879a000ULL, 0x3ff2539d838ff5bdULL, 0xbfc160c8252ca000ULL,
0x3ff23eb7aac9083bULL, 0xbfc0ce7f57f72000ULL, 0x3ff22a012ba940b6ULL, 0xbfc03cdc49fea000ULL,
0x3ff2157996cc4132ULL, 0xbfbf57bdbc4b8000ULL, 0x3ff201201dd2fc9bULL, 0xbfbe370896404000ULL,
0x3ff1ecf4494d480bULL, 0xbfbd17983ef94000ULL, 0x3ff1d8f5528f6569ULL, 0xbfbbf9674ed8a000ULL,
0x3ff1c52311577e7cULL, 0xbfbadc79202f6000ULL, 0x3ff1b17c74cb26e9ULL, 0xbfb9c0c3e7288000ULL,
0x3ff19e010c2c1ab6ULL, 0xbfb8a646b372c000ULL, 0x3ff18ab07bb670bdULL, 0xbfb78d01b3ac0000ULL,
0x3ff1778a25efbcb6ULL, 0xbfb674f145380000ULL, 0x3ff1648d354c31daULL, 0xbfb55e0e6d878000ULL,
0x3ff151b990275fddULL, 0xbfb4485cdea1e000ULL, 0x3ff13f0ea432d24cULL, 0xbfb333d94d6aa000ULL,
0x3ff12c8b7210f9daULL, 0xbfb22079f8c56000ULL, 0x3ff11a3028ecb531ULL, 0xbfb10e4698622000ULL,
0x3ff107fbda8434afULL, 0xbfaffa6c6ad20000ULL, 0x3ff0f5ee0f4e6bb3ULL, 0xbfadda8d4a774000ULL,
0x3ff0e4065d2a9fceULL, 0xbfabbcece4850000ULL, 0x3ff0d244632ca521ULL, 0xbfa9a1894012c000ULL,
0x3ff0c0a77ce2981aULL, 0xbfa788583302c000ULL, 0x3ff0af2f83c636d1ULL, 0xbfa5715e67d68000ULL,
0x3ff09ddb98a01339ULL, 0xbfa35c8a49658000ULL, 0x3ff08cabaf52e7dfULL, 0xbfa149e364154000ULL,
0x3ff07b9f2f4e28fbULL, 0xbf9e72c082eb8000ULL, 0x3ff06ab58c358f19ULL, 0xbf9a55f152528000ULL,
0x3ff059eea5ecf92cULL, 0xbf963d62cf818000ULL, 0x3ff04949cdd12c90ULL, 0xbf9228fb8caa0000ULL,
0x3ff038c6c6f0ada9ULL, 0xbf8c317b20f90000ULL, 0x3ff02865137932a9ULL, 0xbf8419355daa0000ULL,
0x3ff0182427ea7348ULL, 0xbf781203c2ec0000ULL, 0x3ff008040614b195ULL, 0xbf60040979240000ULL,
0x3fefe01ff726fa1aULL, 0x3f6feff384900000ULL, 0x3fefa11cc261ea74ULL, 0x3f87dc41353d0000ULL,
0x3fef6310b081992eULL, 0x3f93cea3c4c28000ULL, 0x3fef25f63ceeadcdULL, 0x3f9b9fc114890000ULL,
0x3feee9c8039113e7ULL, 0x3fa1b0d8ce110000ULL, 0x3feeae8078cbb1abULL, 0x3fa58a5bd001c000ULL,
0x3fee741aa29d0c9bULL, 0x3fa95c8340d88000ULL, 0x3fee3a91830a99b5ULL, 0x3fad276aef578000ULL,
0x3fee01e009609a56ULL, 0x3fb07598e598c000ULL, 0x3fedca01e577bb98ULL, 0x3fb253f5e30d2000ULL,
0x3fed92f20b7c9103ULL, 0x3fb42edd8b380000ULL, 0x3fed5cac66fb5cceULL, 0x3fb606598757c000ULL,
0x3fed272caa5ede9dULL, 0x3fb7da76356a0000ULL, 0x3fecf26e3e6b2ccdULL, 0x3fb9ab434e1c6000ULL,
0x3fecbe6da2a77902ULL, 0x3fbb78c7bb0d6000ULL, 0x3fec8b266d37086dULL, 0x3fbd431332e72000ULL,
0x3fec5894bd5d5804ULL, 0x3fbf0a3171de6000ULL, 0x3fec26b533bb9f8cULL, 0x3fc067152b914000ULL,
0x3febf583eeece73fULL, 0x3fc147858292b000ULL, 0x3febc4fd75db96c1ULL, 0x3fc2266ecdca3000ULL,
0x3feb951e0c864a28ULL, 0x3fc303d7a6c55000ULL, 0x3feb65e2c5ef3e2cULL, 0x3fc3dfc33c331000ULL,
0x3feb374867c9888bULL, 0x3fc4ba366b7a8000ULL, 0x3feb094b211d304aULL, 0x3fc5933928d1f000ULL,
0x3feadbe885f2ef7eULL, 0x3fc66acd2418f000ULL, 0x3feaaf1d31603da2ULL, 0x3fc740f8ec669000ULL,
0x3fea82e63fd358a7ULL, 0x3fc815c0f51af000ULL, 0x3fea5740ef09738bULL, 0x3fc8e92954f68000ULL,
0x3fea2c2a90ab4b27ULL, 0x3fc9bb3602f84000ULL, 0x3fea01a01393f2d1ULL, 0x3fca8bed1c2c0000ULL,
0x3fe9d79f24db3c1bULL, 0x3fcb5b515c01d000ULL, 0x3fe9ae2505c7b190ULL, 0x3fcc2967ccbcc000ULL,
0x3fe9852ef297ce2fULL, 0x3fccf635d5486000ULL, 0x3fe95cbaeea44b75ULL, 0x3fcdc1bd3446c000ULL,
0x3fe934c69de74838ULL, 0x3fce8c01b8cfe000ULL, 0x3fe90d4f2f6752e6ULL, 0x3fcf5509c0179000ULL,
0x3fe8e6528effd79dULL, 0x3fd00e6c121fb800ULL, 0x3fe8bfce9fcc007cULL, 0x3fd071b80e93d000ULL,
0x3fe899c0dabec30eULL, 0x3fd0d46b9e867000ULL, 0x3fe87427aa2317fbULL, 0x3fd13687334bd000ULL,
0x3fe84f00acb39a08ULL, 0x3fd1980d67234800ULL, 0x3fe82a49e8653e55ULL, 0x3fd1f8ffe0cc8000ULL,
0x3fe8060195f40260ULL, 0x3fd2595fd7636800ULL, 0x3fe7e22563e0a329ULL, 0x3fd2b9300914a800ULL,
0x3fe7beb377dcb5adULL, 0x3fd3187210436000ULL, 0x3fe79baa679725c2ULL, 0x3fd377266dec1800ULL,
0x3fe77907f2170657ULL, 0x3fd3d54ffbaf3000ULL, 0x3fe756cadbd6130cULL, 0x3fd432eee32fe000ULL};
  *invc = as_f64(T[2 * i]);
  *logc = as_f64(T[2 * i + 1]);
}

EM_FN double log(double x) {
  uint64_t ix = as_u64(x);
  uint32_t top = static_cast<uint32_t>(ix >> 48);
  if (ix - 0x3fee000000000000ull < 0x3ff1090000000000ull - 0x3fee000000000000ull) {
    // close to 1.0
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double B0 = -0x1.0000000000000p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3,
                 B3 = 0x1.999999995dd0cp-3, B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3,
                 B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4,
                 B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
    const double r = x - 1.0;
    const double r2 = r * r;
    const double r3 = r * r2;
    double p1 = __builtin_fma(r, B2, B1);
    p1 = __builtin_fma(r2, B3, p1);
    double p2 = __builtin_fma(r, B5, B4);
    p2 = __builtin_fma(r2, B6, p2);
    double p3 = __builtin_fma(r, B8, B7);
    p3 = __builtin_fma(r2, B9, p3);
    p3 = __builtin_fma(r3, B10, p3);
    double q = __builtin_fma(p3, r3, p2);
    q = __builtin_fma(q, r3, p1);
    const double t = __builtin_fma(r, 0x1p27, r);
    const double rhi = __builtin_fma(-0x1p27, r, t);
    const double rlo = r - rhi;
    const double sq = rhi * rhi;
    const double hi = __builtin_fma(sq, B0, r);
    double lo = __builtin_fma(sq, B0, r - hi);
    lo = __builtin_fma(B0 * rlo, rhi + r, lo);
    const double y = __builtin_fma(q, r3, lo);
    return hi + y;
  }
  if (top - 0x0010 >= 0x7ff0 - 0x0010) {
    if (ix * 2 == 0) return -1.0 / 0.0;
    if (ix == 0x7ff0000000000000ull) return x;
    if ((top & 0x8000) || (top & 0x7ff0) == 0x7ff0) return (x - x) / (x - x);
    ix = as_u64(x * 0x1p52);  // subnormal: normalise
    ix -= 52ull << 52;
  }
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const uint32_t i = static_cast<uint32_t>(tmp >> 45) & 127;
  const int64_t k = static_cast<int64_t>(tmp) >> 52;
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  double invc, logc;
  log_tab(i, &invc, &logc);
  const double z = as_f64(iz);
  const double r = __builtin_fma(z, invc, -1.0);
  const double kd = static_cast<double>(k);
  const double w = __builtin_fma(kd, 0x1.62e42fefa3800p-1, logc);
  const double hi = r + w;
  double lo = (w - hi) + r;
  lo = __builtin_fma(kd, 0x1.ef35793c76730p-45, lo);
  const double r2 = r * r;
  const double pa = __builtin_fma(r, -0x1.fffffffeb4590p-3, 0x1.555555551305bp-2);
  const double pb = __builtin_fma(r, -0x1.55575e506c89fp-3, 0x1.999b324f10111p-3);
  const double lo2 = __builtin_fma(r2, -0x1.0000000000001p-1, lo);
  const double p = __builtin_fma(pb, r2, pa);
  const double y = __builtin_fma(r * r2, p, lo2);
  return y + hi;
}

EM_FN int32_t hi_word(double x) { return static_cast<int32_t>(as_u64(x) >> 32); }
EM_FN double with_hi_word(double x, uint32_t hi) { return as_f64((as_u64(x) & 0xffffffffull) | (uint64_t(hi) << 32)); }

// glibc 2.35 sysdeps/ieee754/dbl-64/s_log1p.c (fdlibm), as written
EM_FN double log1p_general(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               two54 = 1.80143985094819840000e+16;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  const double zero = 0.0;
  double hfsq, f = 0.0, c = 0.0, s, z, R, u, z2, z4, z6, R1, R2, R3, R4;
  int32_t k, hx, hu = 0, ax;
  hx = hi_word(x);
  ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) {       // x < 0.41422
    if (ax >= 0x3ff00000) {    // x <= -1.0
      if (x == -1.0) return -two54 / zero;
      return (x - x) / (x - x);
    }
    if (ax < 0x3e200000) {     // |x| < 2**-29
      if (ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= static_cast<int32_t>(0xbfd2bec3)) {
      k = 0;
      f = x;
      hu = 1;
    }
  } else if (hx >= 0x7ff00000) {
    return x + x;
  }
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = hi_word(u);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
      c /= u;
    } else {
      u = x;
      hu = hi_word(u);
      k = (hu >> 20) - 1023;
      c = 0;
    }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = with_hi_word(u, static_cast<uint32_t>(hu | 0x3ff00000));
    } else {
      k += 1;
      u = with_hi_word(u, static_cast<uint32_t>(hu | 0x3fe00000));
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) {  // |f| < 2**-20
    if (f == zero) {
      if (k == 0) return zero;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  s = f / (2.0 + f);
  z = s * s;
  R1 = z * Lp1;
  z2 = z * z;
  R2 = Lp2 + z * Lp3;
  z4 = z2 * z2;
  R3 = Lp4 + z * Lp5;
  z6 = z4 * z2;
  R4 = Lp6 + z * Lp7;
  R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// The same function with the common range in select form, exactly as the float log1pf above (whose
// restructuring is checked against glibc on every float): x finite, -1 < x < 2^53, |x| >= 2^-29, reduced
// argument not within 2^-20 of a power of two; everything else goes to log1p_general.
EM_FN double log1p(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  const int32_t hx = hi_word(x);
  const int32_t ax = hx & 0x7fffffff;
  if ((hx < 0 && ax >= 0x3ff00000) || ax < 0x3e200000 || hx >= 0x43400000) return log1p_general(x);
  double f, c = 0.0;
  int32_t k = 0;
  if (hx < 0x3FDA827A && (hx > 0 || hx <= static_cast<int32_t>(0xbfd2bec3))) {
    f = x;  // -0.2929 < x < 0.41422
  } else {
    double u = 1.0 + x;
    int32_t hu = hi_word(u);
    k = (hu >> 20) - 1023;
    c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
    c /= u;
    hu &= 0x000fffff;
    const bool low = hu < 0x6a09e;
    k += low ? 0 : 1;
    u = with_hi_word(u, static_cast<uint32_t>(hu | (low ? 0x3ff00000 : 0x3fe00000)));
    hu = low ? hu : ((0x00100000 - hu) >> 2);
    if (hu == 0) return log1p_general(x);  // |f| < 2^-20
    f = u - 1.0;
  }
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double R1 = z * Lp1;
  const double z2 = z * z;
  const double R2 = Lp2 + z * Lp3;
  const double z4 = z2 * z2;
  const double R3 = Lp4 + z * Lp5;
  const double z6 = z4 * z2;
  const double R4 = Lp6 + z * Lp7;
  const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  const double kd = static_cast<double>(k);
  const double r0 = f - (hfsq - s * (hfsq + R));
  const double rk = kd * ln2_hi - ((hfsq - (s * (hfsq + R) + (kd * ln2_lo + c))) - f);
  return (k == 0) ? r0 : rk;
}

// glibc 2.35 sysdeps/ieee754/dbl-64/s_expm1.c (fdlibm), as written
EM_FN double expm1_general(double x) {
  const double one = 1.0, huge = 1.0e+300, tiny = 1.0e-300;
  const double o_threshold = 7.09782712893383973096e+02, ln2_hi = 6.93147180369123816490e-01,
               ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
               Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
               Q5 = -2.01099218183624371326e-07;
  double y, hi, lo, c = 0.0, t, e, hxs, hfx, r1, h2, h4, R1, R2, R3;
  int32_t k, xsb;
  uint32_t hx = static_cast<uint32_t>(hi_word(x));
  xsb = static_cast<int32_t>(hx & 0x80000000u);
  hx &= 0x7fffffffu;
  if (hx >= 0x4043687Au) {      // |x| >= 56 ln2
    if (hx >= 0x40862E42u) {    // |x| >= 709.78
      if (hx >= 0x7ff00000u) {
        const uint32_t low = static_cast<uint32_t>(as_u64(x));
        if (((hx & 0xfffff) | low) != 0) return x + x;
        return (xsb == 0) ? x : -1.0;
      }
      if (x > o_threshold) return huge * huge;
    }
    if (xsb != 0) return tiny - one;
  }
  if (hx > 0x3fd62e42u) {    // |x| > 0.5 ln2
    if (hx < 0x3FF0A2B2u) {  // |x| < 1.5 ln2
      if (xsb == 0) {
        hi = x - ln2_hi;
        lo = ln2_lo;
        k = 1;
      } else {
        hi = x + ln2_hi;
        lo = -ln2_lo;
        k = -1;
      }
    } else {
      k = static_cast<int32_t>(invln2 * x + ((xsb == 0) ? 0.5 : -0.5));
      t = k;
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x3c900000u) {  // |x| < 2**-54
    t = huge + x;
    return x - (t - (huge + x));
  } else {
    k = 0;
  }
  hfx = 0.5 * x;
  hxs = x * hfx;
  R1 = one + hxs * Q1;
  h2 = hxs * hxs;
  R2 = Q2 + hxs * Q3;
  h4 = h2 * h2;
  R3 = Q4 + hxs * Q5;
  r1 = R1 + h2 * R2 + h4 * R3;
  t = 3.0 - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1) {
    if (x < -0.25) return -2.0 * (e - (x + 0.5));
    return one + 2.0 * (x - e);
  }
  if (k <= -2 || k > 56) {
    y = one - (e - x);
    y = with_hi_word(y, static_cast<uint32_t>(hi_word(y)) + (static_cast<uint32_t>(k) << 20));
    return y - one;
  }
  t = one;
  if (k < 20) {
    t = with_hi_word(t, 0x3ff00000u - (0x200000u >> k));  // 1 - 2^-k
    y = t - (e - x);
    y = with_hi_word(y, static_cast<uint32_t>(hi_word(y)) + (static_cast<uint32_t>(k) << 20));
  } else {
    t = with_hi_word(t, static_cast<uint32_t>(0x3ff - k) << 20);  // 2^-k
    y = x - (e + t);
    y += one;
    y = with_hi_word(y, static_cast<uint32_t>(hi_word(y)) + (static_cast<uint32_t>(k) << 20));
  }
  return y;
}

// The same function with the common range (2^-54 <= |x| < 56 ln 2) in select form, exactly as the float
// expm1f above (whose restructuring is checked against glibc on every float): one reduction formula
// (k = 0 and k = +-1 are the general formula with t = 0 and t = +-1), reconstructions selected from
// straight-line candidates; everything else goes to expm1_general.
EM_FN double expm1(double x) {
  const double one = 1.0, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
               Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
               Q5 = -2.01099218183624371326e-07;
  const uint32_t hw = static_cast<uint32_t>(hi_word(x));
  const uint32_t hx = hw & 0x7fffffffu;
  if (hx >= 0x4043687Au || hx < 0x3c900000u) return expm1_general(x);
  const bool neg = (hw >> 31) != 0;
  double kd = 0.0;
  if (hx > 0x3fd62e42u) {
    const double general = static_cast<double>(static_cast<int32_t>(invln2 * x + (neg ? -0.5 : 0.5)));
    kd = (hx < 0x3FF0A2B2u) ? (neg ? -1.0 : 1.0) : general;
  }
  const int32_t k = static_cast<int32_t>(kd);
  const double hi = x - kd * ln2_hi;
  const double lo = kd * ln2_lo;
  const double xr = hi - lo;
  const double c = (hi - xr) - lo;
  const double hfx = 0.5 * xr;
  const double hxs = xr * hfx;
  const double R1 = one + hxs * Q1;
  const double h2 = hxs * hxs;
  const double R2 = Q2 + hxs * Q3;
  const double h4 = h2 * h2;
  const double R3 = Q4 + hxs * Q5;
  const double r1 = R1 + h2 * R2 + h4 * R3;
  const double t = 3.0 - r1 * hfx;
  const double e = hxs * ((r1 - t) / (6.0 - xr * t));
  const double r0 = xr - (xr * e - hxs);                                  // k == 0
  const double e2 = (xr * (e - c) - c) - hxs;
  const double rm1 = 0.5 * (xr - e2) - 0.5;                               // k == -1
  const double rp1 = (xr < -0.25) ? -2.0 * (e2 - (xr + 0.5)) : one + 2.0 * (xr - e2);  // k == 1
  const double dd = e2 - xr;
  const uint32_t kbits = static_cast<uint32_t>(k) << 20;
  auto scaled = [&](double y) { return with_hi_word(y, static_cast<uint32_t>(hi_word(y)) + kbits); };
  const double ya = scaled(one - dd) - one;                               // k <= -2 (k > 56 is outside this range)
  const uint32_t ksmall = (k >= 2 && k < 20) ? static_cast<uint32_t>(k) : 2u;
  const double t1 = with_hi_word(one, 0x3ff00000u - (0x200000u >> ksmall));  // 1 - 2^-k
  const double yb = scaled(t1 - dd);                                      // 2 <= k < 20
  const uint32_t klarge = (k >= 20 && k <= 1022) ? static_cast<uint32_t>(k) : 20u;
  const double t2 = with_hi_word(one, (0x3ffu - klarge) << 20);           // 2^-k
  const double yc = scaled((xr - (e2 + t2)) + one);                       // k >= 20
  double r = (k < 20) ? yb : yc;
  r = (k <= -2) ? ya : r;
  r = (k == 1) ? rp1 : r;
  r = (k == -1) ? rm1 : r;
  r = (k == 0) ? r0 : r;
  return r;
}

// glibc 2.35 sysdeps/ieee754/dbl-64/s_tanh.c (fdlibm)
EM_FN double tanh(double x) {
  const double one = 1.0, two = 2.0, tiny = 1.0e-300;
  double t, z;
  const int32_t jx = hi_word(x);
  const uint32_t lx = static_cast<uint32_t>(as_u64(x));
  const int32_t ix = jx & 0x7fffffff;
  if (ix >= 0x7ff00000) {
    if (jx >= 0) return one / x + one;
    return one / x - one;
  }
  if (ix < 0x40360000) {  // |x| < 22
    if ((static_cast<uint32_t>(ix) | lx) == 0) return x;
    if (ix < 0x3c800000) return x * (one + x);  // |x| < 2**-55
    // one expm1 evaluation and one division for both ranges of the source (|x| >= 1:
    // 1 - 2/(expm1(2|x|) + 2); |x| < 1: -t/(t + 2), t = expm1(-2|x|)): per lane the same operations
    // on the same values (see the float tanhf)
    const double ax = as_f64(as_u64(x) & 0x7fffffffffffffffull);
    const bool big = ix >= 0x3ff00000;  // |x| >= 1
    t = expm1(big ? two * ax : -two * ax);
    const double q = (big ? two : -t) / (t + two);
    z = big ? one - q : q;
  } else {
    z = one - tiny;
  }
  return (jx >= 0) ? z : -z;
}

// The Phi rule's function in double precision, -ln(tanh(max(x, 1e-30) / 2)) (arithmetic.rs:180-186): tanh, expm1 and log above,
// fused for this argument as phif is for f32 (round 6: Phif64 is the reference CLI's default decoder, src/cli/ber.rs:48-50).
// The argument of tanh is h >= 5e-31 (or +inf; a NaN x becomes 1e-30 in the max): its sign and NaN classes drop out, the
// argument of expm1 is 2h in [2, 44) or -2h in (-2, -2^-54], so expm1's overflow, NaN, k = 1 and tiny classes cannot occur
// (k is 3..63 or -3..0; k > 56 -- h >= 19.4, which the select form of expm1 above leaves to expm1_general -- takes the
// same formula as k <= -2); tanh's result is a normal number in (0, 1], so log's zero / negative / infinite / NaN /
// subnormal classes cannot occur either.  One straight line of selects: a wavefront whose lanes sit in different classes
// (|h| < 1 and >= 1, log's "close to 1" polynomial and its table path) runs no divergent branches.  Per lane the
// operations are the three functions'; tools/check_exact_math64.cpp compares it with them and with glibc.
EM_FN double phi(double x) {
  const double one = 1.0, two = 2.0, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
               Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
               Q5 = -2.01099218183624371326e-07;
  const double h = 0.5 * fmax(x, 1e-30);
  // tanh(h), h > 0
  const uint32_t ix = static_cast<uint32_t>(hi_word(h));
  const bool sat = ix >= 0x40360000u;                  // h >= 22, infinity: one - tiny = 1
  const bool tiny = ix < 0x3c800000u;                  // h < 2^-55: h * (1 + h)
  const double ax = (sat || tiny) ? one : h;           // (any ordinary argument for the lanes whose result is replaced)
  const bool big = ix >= 0x3ff00000u;                  // h >= 1
  const double arg = big ? two * ax : -two * ax;
  const uint32_t hx = static_cast<uint32_t>(hi_word(arg)) & 0x7fffffffu;
  const double general = static_cast<double>(static_cast<int32_t>(invln2 * arg + (big ? 0.5 : -0.5)));
  double kd = (hx < 0x3FF0A2B2u) ? -1.0 : general;     // (|arg| < 1.5 ln2 happens only for the negative arguments)
  kd = (hx > 0x3fd62e42u) ? kd : 0.0;
  const int32_t k = static_cast<int32_t>(kd);
  const double hi = arg - kd * ln2_hi;
  const double lo = kd * ln2_lo;
  const double xr = hi - lo;
  const double c = (hi - xr) - lo;
  const double hfx = 0.5 * xr;
  const double hxs = xr * hfx;
  const double R1 = one + hxs * Q1;
  const double h2 = hxs * hxs;
  const double R2 = Q2 + hxs * Q3;
  const double h4 = h2 * h2;
  const double R3 = Q4 + hxs * Q5;
  const double r1 = R1 + h2 * R2 + h4 * R3;
  const double t3 = 3.0 - r1 * hfx;
  const double e = hxs * ((r1 - t3) / (6.0 - xr * t3));
  const double r0 = xr - (xr * e - hxs);                                  // k == 0
  const double e2 = (xr * (e - c) - c) - hxs;
  const double rm1 = 0.5 * (xr - e2) - 0.5;                               // k == -1
  const double dd = e2 - xr;
  const uint32_t kbits = static_cast<uint32_t>(k) << 20;
  auto scaled = [&](double y) { return with_hi_word(y, static_cast<uint32_t>(hi_word(y)) + kbits); };
  const double ya = scaled(one - dd) - one;                               // k <= -2, k > 56
  const uint32_t ksmall = (k >= 2 && k < 20) ? static_cast<uint32_t>(k) : 2u;
  const double t1 = with_hi_word(one, 0x3ff00000u - (0x200000u >> ksmall));  // 1 - 2^-k
  const double yb = scaled(t1 - dd);                                      // 2 <= k < 20
  const uint32_t klarge = (k >= 20 && k <= 1022) ? static_cast<uint32_t>(k) : 20u;
  const double t2 = with_hi_word(one, (0x3ffu - klarge) << 20);           // 2^-k
  const double yc = scaled((xr - (e2 + t2)) + one);                       // 20 <= k <= 56
  double t = (k < 20) ? yb : yc;
  t = (k <= -2 || k > 56) ? ya : t;
  t = (k == -1) ? rm1 : t;
  t = (k == 0) ? r0 : t;
  const double q = (big ? two : -t) / (t + two);
  double z = big ? one - q : q;
  z = sat ? one : z;
  z = tiny ? h * (one + h) : z;
  // log(z), z normal in (0, 1]
  const uint64_t iz0 = as_u64(z);
  const bool near1 = iz0 - 0x3fee000000000000ull < 0x3ff1090000000000ull - 0x3fee000000000000ull;
  double lg_near;
  {
    const double B0 = -0x1.0000000000000p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3,
                 B3 = 0x1.999999995dd0cp-3, B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3,
                 B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4,
                 B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
    const double r = z - 1.0;
    const double r2 = r * r;
    const double r3 = r * r2;
    double p1 = __builtin_fma(r, B2, B1);
    p1 = __builtin_fma(r2, B3, p1);
    double p2 = __builtin_fma(r, B5, B4);
    p2 = __builtin_fma(r2, B6, p2);
    double p3 = __builtin_fma(r, B8, B7);
    p3 = __builtin_fma(r2, B9, p3);
    p3 = __builtin_fma(r3, B10, p3);
    double qq = __builtin_fma(p3, r3, p2);
    qq = __builtin_fma(qq, r3, p1);
    const double tt = __builtin_fma(r, 0x1p27, r);
    const double rhi = __builtin_fma(-0x1p27, r, tt);
    const double rlo = r - rhi;
    const double sq = rhi * rhi;
    const double hi1 = __builtin_fma(sq, B0, r);
    double lo1 = __builtin_fma(sq, B0, r - hi1);
    lo1 = __builtin_fma(B0 * rlo, rhi + r, lo1);
    const double y = __builtin_fma(qq, r3, lo1);
    lg_near = (iz0 == 0x3ff0000000000000ull) ? 0.0 : hi1 + y;
  }
  double lg_far;
  {
    const uint64_t tmp = iz0 - 0x3fe6000000000000ull;
    const uint32_t i = static_cast<uint32_t>(tmp >> 45) & 127;
    const int64_t kk = static_cast<int64_t>(tmp) >> 52;
    const uint64_t iz = iz0 - (tmp & (0xfffull << 52));
    double invc, logc;
    log_tab(i, &invc, &logc);
    const double zz = as_f64(iz);
    const double r = __builtin_fma(zz, invc, -1.0);
    const double kdd = static_cast<double>(kk);
    const double w = __builtin_fma(kdd, 0x1.62e42fefa3800p-1, logc);
    const double hi2 = r + w;
    double lo2 = (w - hi2) + r;
    lo2 = __builtin_fma(kdd, 0x1.ef35793c76730p-45, lo2);
    const double r2 = r * r;
    const double pa = __builtin_fma(r, -0x1.fffffffeb4590p-3, 0x1.555555551305bp-2);
    const double pb = __builtin_fma(r, -0x1.55575e506c89fp-3, 0x1.999b324f10111p-3);
    const double lo3 = __builtin_fma(r2, -0x1.0000000000001p-1, lo2);
    const double pp = __builtin_fma(pb, r2, pa);
    const double y = __builtin_fma(r * r2, pp, lo3);
    lg_far = y + hi2;
  }
  const double lg = near1 ? lg_near : lg_far;
  return (h != h) ? h : -lg;
}

}  // namespace em
}  // namespace ldpc
