// sk_lean_math.h -- log / exp / x^y for the per-source strengths of the spread kernels, written so that every
// polynomial constant is the MULTIPLICAND of its fused multiply-add (v_fma_f64 acc, s[const], v[power]).  The library
// versions evaluate Horner chains whose constants are addends; the compiler materialises each of those in a VGPR pair
// and, the constants being loop invariant, keeps them alive across the whole tile loop: 60 registers more, which is the
// difference between four and three workgroups per CU for k_spread_b (or spills).  Accuracy (checked against libm on
// the host by tests/test_host_logic.py through sk_host_lean_math): log <= 1 ulp, exp <= 1 ulp on its reduced range,
// pow_lean(x, y) relative error <= |y log x| x 2.3e-16, i.e. <= 2e-14 for the exponents (|y| <= 4) and arguments of this
// path -- two orders below the accuracy of the NUFFT that consumes the strengths.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

#if defined(__HIPCC__)
#define SK_HD __host__ __device__ __forceinline__
#else
#define SK_HD inline
#endif

namespace sk {

SK_HD int64_t lean_bits(double x) {
  int64_t b;
  memcpy(&b, &x, 8);
  return b;
}
SK_HD double lean_from_bits(int64_t b) {
  double x;
  memcpy(&x, &b, 8);
  return x;
}

// natural logarithm of a finite x > 0 (any other argument: the caller's business, see pow_lean)
SK_HD double log_lean(double x) {
  int64_t b = lean_bits(x);
  int k = 0;
  if (b < (int64_t)0x0010000000000000LL) {  // subnormal: scale up
    x *= 18014398509481984.0;               // 2^54
    b = lean_bits(x);
    k = -54;
  }
  // x = 2^k m, m in [sqrt(1/2), sqrt(2))
  int hx = (int)(b >> 32);
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int i = (hx + 0x95f64) & 0x100000;
  k += i >> 20;
  const double m = lean_from_bits(((int64_t)(hx | (i ^ 0x3ff00000)) << 32) | (b & 0xffffffffLL));
  const double f = m - 1.0;
  // s = f / (2 + f) by a reciprocal seed and two Newton steps on the quotient (correct to < 1 ulp)
  const double d = 2.0 + f;
#if defined(__HIP_DEVICE_COMPILE__)
  double rc = __builtin_amdgcn_rcp(d);
#else
  double rc = 1.0 / d;
#endif
  rc = fma(fma(-d, rc, 1.0), rc, rc);
  double s = f * rc;
  s = fma(fma(-d, s, f), rc, s);
  const double z = s * s, w = z * z;
  // fdlibm's minimax coefficients Lg1..Lg7 of (log(1+s) - log(1-s))/s - 2 in z = s^2, summed against powers
  const double w2 = w * w, zw = z * w;
  double R = 1.479819860511658591e-01 * (zw * w2);      // Lg7 z^7
  R = fma(1.531383769920937332e-01, w * w2, R);         // Lg6 z^6
  R = fma(1.818357216161805012e-01, z * w2, R);         // Lg5 z^5
  R = fma(2.222219843214978396e-01, w2, R);             // Lg4 z^4
  R = fma(2.857142874366239149e-01, zw, R);             // Lg3 z^3
  R = fma(3.999999999940941908e-01, w, R);              // Lg2 z^2
  R = fma(6.666666666666735130e-01, z, R);              // Lg1 z
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  // log x = k ln2 + f - hfsq + s (hfsq + R), with ln2 split so that k ln2_hi is exact
  return fma(dk, 6.93147180369123816490e-01, f - (hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)));
}

// e^z for any finite z (overflow -> inf, underflow -> 0)
SK_HD double exp_lean(double z) {
  if (!(z > -745.2)) return z == z ? 0.0 : z;
  if (z > 709.8) return INFINITY;
  const double kd = rint(z * 1.44269504088896338700e+00);
  double r = fma(-kd, 6.93147180369123816490e-01, z);
  r = fma(-kd, 1.90821492927058770002e-10, r);  // |r| <= 0.3466
  // Taylor polynomial of degree 13 (truncation 4e-18 relative), as A(r) + r^4 B(r) + r^8 C(r) + r^12 (c12 + c13 r),
  // every block summed against r, r^2, r^3 with its constants as multiplicands
  const double r2 = r * r, r3 = r2 * r, r4 = r2 * r2, r8 = r4 * r4;
  double A = 1.66666666666666657415e-01 * r3;
  A = fma(0.5, r2, A);
  A += r;
  double B = 1.98412698412698412526e-04 * r3;            // 1/7!
  B = fma(1.38888888888888894189e-03, r2, B);             // 1/6!
  B = fma(8.33333333333333321769e-03, r, B);              // 1/5!
  B += 4.16666666666666643537e-02;                        // 1/4!
  double C = 2.50521083854417202239e-08 * r3;             // 1/11!
  C = fma(2.75573192239858882545e-07, r2, C);             // 1/10!
  C = fma(2.75573192239858925110e-06, r, C);              // 1/9!
  C += 2.48015873015873015658e-05;                        // 1/8!
  const double D = fma(1.60590438368216133250e-10, r, 2.08767569878680981860e-09);  // 1/13!, 1/12!
  double p = fma(r8, fma(r4, D, C), fma(r4, B, A));       // e^r - 1
  // 2^k (1 + p): k in [-1075, 1024] -- scale in two exact steps so that subnormal results round once
  const int k = (int)kd;
  const int k1 = k / 2, k2 = k - k1;
  const double s1 = lean_from_bits((int64_t)(k1 + 1023) << 52), s2 = lean_from_bits((int64_t)(k2 + 1023) << 52);
  return (fma(p, s1, s1)) * s2;
}

// x^y for x >= 0 (x = 0: 0, 1 or inf by the sign of y, as pow; NaN propagates)
SK_HD double pow_lean(double x, double y) {
  if (!(x > 0.0)) return x == 0.0 ? (y > 0.0 ? 0.0 : (y == 0.0 ? 1.0 : INFINITY)) : NAN;
  if (x > 1.7e308) return y > 0.0 ? INFINITY : (y == 0.0 ? 1.0 : 0.0);
  return exp_lean(y * log_lean(x));
}

}  // namespace sk
