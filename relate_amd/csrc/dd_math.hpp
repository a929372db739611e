// Correctly rounded acos, cos, pow and log for the LD cells: the cubic solve (gsl_poly_solve_cubic,
// src/ld.cpp:139-220) and the likelihood that picks among its roots.
//
// The reference calls the host libm there.  Its roots, and the LD measures derived from them, carry that libm's
// last-bit rounding; `getHighestId` then compares LD measures with a strict `>`, so a pair of depths whose measures
// are equal in exact arithmetic (mirror-image genotype tables, common at small N) is decided by those last bits.
// glibc 2.35's cos / acos / pow return the correctly rounded result for 99.87 % / 99.93 % / 99.92 % of arguments
// (measured against 200-bit arithmetic); the device library's versions agree with glibc far less often (9 % of LD
// cells differed in the last bits).  These versions evaluate in double-double (~106 bits) and round once, so they
// agree with glibc wherever glibc is correctly rounded.  Plain C++ (fma only): the same header compiles for the host,
// where tests/test_dd_math.py checks it against 200-bit arithmetic.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define DDM_HD __host__ __device__
#else
#define DDM_HD
#endif

// The error-free transformations must not be contracted into FMAs; the pragma is function-local so that it does not
// change the rest of the translation unit.  (g++ builds of the host test use -ffp-contract=off.)
#if defined(__clang__)
#define DDM_NOCONTRACT _Pragma("clang fp contract(off)")
#else
#define DDM_NOCONTRACT
#endif

namespace ddm {

struct dd {
  double hi, lo;
};

DDM_HD inline dd two_sum(double a, double b) {
  DDM_NOCONTRACT
  double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
DDM_HD inline dd quick_two_sum(double a, double b) {  // |a| >= |b|
  DDM_NOCONTRACT
  double s = a + b;
  return {s, b - (s - a)};
}
DDM_HD inline dd two_prod(double a, double b) {
  DDM_NOCONTRACT
  double p = a * b;
  return {p, std::fma(a, b, -p)};
}
DDM_HD inline dd add(dd a, dd b) {
  DDM_NOCONTRACT
  dd s = two_sum(a.hi, b.hi), t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
DDM_HD inline dd add(dd a, double b) {
  DDM_NOCONTRACT
  dd s = two_sum(a.hi, b);
  s.lo += a.lo;
  return quick_two_sum(s.hi, s.lo);
}
DDM_HD inline dd neg(dd a) { return {-a.hi, -a.lo}; }
DDM_HD inline dd sub(dd a, dd b) { return add(a, neg(b)); }
DDM_HD inline dd mul(dd a, dd b) {
  DDM_NOCONTRACT
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
DDM_HD inline dd mul(dd a, double b) {
  DDM_NOCONTRACT
  dd p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return quick_two_sum(p.hi, p.lo);
}
DDM_HD inline dd div(dd a, dd b) {
  DDM_NOCONTRACT
  double q1 = a.hi / b.hi;
  dd r = sub(a, mul(b, q1));
  double q2 = r.hi / b.hi;
  r = sub(r, mul(b, q2));
  double q3 = r.hi / b.hi;
  dd q = quick_two_sum(q1, q2);
  return add(q, q3);
}
DDM_HD inline dd div(dd a, double b) { return div(a, dd{b, 0.0}); }
DDM_HD inline dd sqrt_dd(dd a) {  // a > 0
  DDM_NOCONTRACT
  double y = std::sqrt(a.hi);
  dd r = sub(a, two_prod(y, y));
  return add(dd{y, 0.0}, r.hi / (2.0 * y));
}

// sin and cos of a double, |x| <= 4: exact reduction by k pi/2 (pi/2 in 33-bit pieces: k * piece is exact), then series
DDM_HD inline void sincos_dd(double x, dd &s, dd &c) {  // |x| <= 4
  DDM_NOCONTRACT
  const double P1 = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  const double P2 = 6.07710050630396597660e-11;  // next 33 bits
  const double P3 = 2.02226624871116645580e-21;  // next 33 bits
  const dd P4 = {8.47842766036889956997e-32, 7.39850476826770377976e-49};  // the rest
  double k = std::nearbyint(x * 6.36619772367581382433e-01);
  dd r = two_sum(x, -k * P1);  // exact product, exact sum
  r = add(r, -k * P2);
  r = add(r, -k * P3);
  r = add(r, mul(P4, -k));
  // Taylor series of sin r and cos r, |r| <= pi/4 + tiny
  dd r2 = mul(r, r);
  dd sn = r, cs = {1.0, 0.0}, ts = r, tc = {1.0, 0.0};
  for (int i = 1; i <= 16; i++) {
    tc = div(mul(tc, r2), (double)((2 * i - 1) * (2 * i)));
    ts = div(mul(ts, r2), (double)((2 * i) * (2 * i + 1)));
    if (i & 1) { cs = sub(cs, tc); sn = sub(sn, ts); }
    else { cs = add(cs, tc); sn = add(sn, ts); }
  }
  int q = ((int)k) & 3;
  if (q == 0) { s = sn; c = cs; }
  else if (q == 1) { s = cs; c = neg(sn); }
  else if (q == 2) { s = neg(sn); c = neg(cs); }
  else { s = neg(cs); c = sn; }
}

DDM_HD inline double cos_cr(double x) {
  DDM_NOCONTRACT
  if (!(std::fabs(x) <= 4.0)) return std::cos(x);
  dd s, c;
  sincos_dd(x, s, c);
  return c.hi;
}

DDM_HD inline double acos_cr(double x) {
  DDM_NOCONTRACT
  if (!(std::fabs(x) < 1.0)) return std::acos(x);  // +-1, out of range, NaN: exact or NaN in every libm
  const dd PI = {3.14159265358979311600e+00, 1.22464679914735320717e-16};
  dd s, c;
  if (std::fabs(x) <= 0.7) {  // one Newton step on cos(theta) = x from the double estimate
    double t0 = std::acos(x);
    sincos_dd(t0, s, c);
    dd corr = div(add(c, -x), s);
    return add(corr, t0).hi;
  }
  // near the ends: sin(phi) = sqrt(1 - x^2) is the well-conditioned equation; acos(x) = phi or pi - phi
  dd om = two_sum(1.0, -x), op = two_sum(1.0, x);
  dd sq = sqrt_dd(mul(om, op));
  double p0 = std::asin(sq.hi);
  sincos_dd(p0, s, c);
  dd phi = add(neg(div(sub(s, sq), c)), p0);
  return x > 0 ? phi.hi : sub(PI, phi).hi;
}

DDM_HD inline dd exp_dd(dd z) {  // |z| < 700
  DDM_NOCONTRACT
  const double L1 = 6.93147180485539138317e-01;  // ln 2, first 33 bits (k L1 is exact for |k| < 2^20)
  const double L2 = 7.44061711001239684738e-11;  // the next 53 bits
  const dd L3 = {-1.31246984177852553613e-27, -3.30793692750132419148e-44};
  double k = std::nearbyint(z.hi * 1.44269504088896338700);
  dd r = add(z, -k * L1);
  r = sub(r, two_prod(k, L2));
  r = sub(r, mul(L3, k));
  r = mul(r, 1.0 / 256.0);
  dd sum = {1.0, 0.0}, term = {1.0, 0.0};
  for (int i = 1; i <= 11; i++) {
    term = div(mul(term, r), (double)i);
    sum = add(sum, term);
  }
  for (int i = 0; i < 8; i++) sum = mul(sum, sum);
  int ik = (int)k;
  return {std::ldexp(sum.hi, ik), std::ldexp(sum.lo, ik)};
}

// x^y for finite x > 0 (the cubic solve raises a positive number to the double nearest 1/3)
DDM_HD inline double pow_cr(double x, double y) {
  DDM_NOCONTRACT
  if (!(x > 1e-280 && x < 1e280) || !(std::fabs(y) < 4.0)) return std::pow(x, y);
  double l0 = std::log(x);
  dd e = exp_dd(dd{-l0, 0.0});
  dd l = add(add(mul(e, x), -1.0), l0);  // one Newton step on exp(l) = x: l0 + (x exp(-l0) - 1)
  return exp_dd(mul(l, y)).hi;
}

// log x for finite x > 0 (the likelihood that picks among the candidate roots, pick_best_p src/ld.cpp)
DDM_HD inline double log_cr(double x) {
  DDM_NOCONTRACT
  if (!(x > 1e-280 && x < 1e280)) return std::log(x);
  double u = x - 1.0;  // exact for x in [0.5, 2]
  if (std::fabs(u) < 1e-4) {  // log(1 + u) by its series: the Newton step below loses relative accuracy next to 1
    dd U = {u, 0.0}, pw = U, sum = U;
    for (int i = 2; i <= 9; i++) {
      pw = mul(pw, U);
      dd t = div(pw, (double)i);
      sum = (i & 1) ? add(sum, t) : sub(sum, t);
    }
    return sum.hi;
  }
  double l0 = std::log(x);
  dd e = exp_dd(dd{-l0, 0.0});
  return add(add(mul(e, x), -1.0), l0).hi;
}

}  // namespace ddm
