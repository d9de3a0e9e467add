// Single-precision exp / log / log1p / tanh that return, bit for bit, what glibc 2.35's libm
// returns on x86-64 -- the functions Rust's f32::{exp, ln, ln_1p, tanh} resolve to on
// linux-gnu, i.e. the ones the reference's transcendental decoder rules are built on
// (/root/reference/src/decoder/arithmetic.rs:184, 357, 376, 510, 965-966).  ocml's versions
// differ in the last ulp, which the ill-conditioned rules (phi's clamp cliff, A-Min*'s argmin
// ties) occasionally amplify; with these the HIP path reproduces the CPU results exactly.
//
// The algorithms are the published ones glibc uses:
//   expf, logf      Szabolcs Nagy's table-driven routines (ARM optimized-routines), evaluated
//                   in double precision; the x86-64 build that runs on FMA-capable CPUs fuses
//                   specific multiply-adds, reproduced here with explicit fma() calls
//   log1pf, expm1f, tanhf   the Sun fdlibm float routines
// Every constant was checked against the libm binary, and tests/ compares each function with
// the host libm exhaustively over all 2^32 arguments (tools/check_exact_math.cpp).
// Works as host code (for that test) and as HIP device code; needs -ffp-contract=off.
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

// glibc 2.35 sysdeps/ieee754/flt-32/s_log1pf.c (fdlibm)
EM_FN float log1pf(float x) {
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

// glibc 2.35 sysdeps/ieee754/flt-32/s_expm1f.c (fdlibm)
EM_FN float expm1f(float x) {
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

// glibc 2.35 sysdeps/ieee754/flt-32/s_tanhf.c (fdlibm)
EM_FN float tanhf(float x) {
  const float one = 1.0f, two = 2.0f, tiny = 1.0e-30f;
  float t, z;
  const int32_t jx = static_cast<int32_t>(as_u32(x));
  const int32_t ix = jx & 0x7fffffff;
  if (ix >= 0x7f800000) {
    if (jx >= 0) return one / x + one;
    return one / x - one;
  }
  if (ix < 0x41b00000) {  // |x| < 22
    if (ix == 0) return x;
    if (ix < 0x24000000) return x * (one + x);  // |x| < 2**-55
    const float ax = as_f32(static_cast<uint32_t>(ix));
    if (ix >= 0x3f800000) {  // |x| >= 1
      t = expm1f(two * ax);
      z = one - two / (t + two);
    } else {
      t = expm1f(-two * ax);
      z = -t / (t + two);
    }
  } else {
    z = one - tiny;
  }
  return (jx >= 0) ? z : -z;
}

}  // namespace em
}  // namespace ldpc
