// amhip_atan_cr.h -- atan(r), r >= 0, CORRECTLY ROUNDED (host + device).
//
// Why: aslam's equidistant distortion calls libm's atan once per (cell, frame) pair
// (ortho-backward-grid.cc:157-171 -> project3 -> distort), and the result decides discrete
// outcomes (image box, rounded keypoint).  The device library's atan and the host's differ by an
// ulp in a good share of their calls; glibc 2.35's is the correctly rounded value in 99.9 % of
// its calls (measured against mpmath: 195 of 200 000 differ).  A correctly rounded device atan
// therefore agrees with the host's in all but those calls -- and owes nothing to either library.
//
// How: double-double arithmetic (~104 bits).
//   r > 1        atan r = pi/2 - atan(1 / r)
//   t in [0, 1]  c = j / 32 nearest to t, y = (t - c) / (1 + t c), |y| <= 1/64:
//                atan t = atan c (table, double-double) + atan y
//   atan y       y (1 + y^2 (-1/3 + y^2 (1/5 - ... + y^2 / 19))): the first dropped term is
//                y^20 / 21 < 2^-124 relative
// The double-double result is rounded to double once.  That is the correctly rounded value
// unless the exact atan lies within ~2^-100 (relative) of a rounding boundary of the doubles.
// tests/test_atan_cr.py compiles this header on the host and compares with mpmath.
#ifndef AMHIP_ATAN_CR_H_
#define AMHIP_ATAN_CR_H_

#include <cmath>

#include "amhip_atan_table.h"

#if defined(__HIPCC__)
#define AMHIP_ATAN_HD __host__ __device__ __forceinline__
#else
#define AMHIP_ATAN_HD inline
#endif

namespace amhip {

struct DD {
  double hi, lo;
};

AMHIP_ATAN_HD DD dd_two_sum(double a, double b) {  // hi + lo == a + b exactly
  const double s = a + b;
  const double bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
AMHIP_ATAN_HD DD dd_fast_two_sum(double a, double b) {  // |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
AMHIP_ATAN_HD DD dd_two_prod(double a, double b) {  // hi + lo == a * b exactly
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
AMHIP_ATAN_HD DD dd_add(const DD& a, const DD& b) {
  DD s = dd_two_sum(a.hi, b.hi);
  const DD t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = dd_fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return dd_fast_two_sum(s.hi, s.lo);
}
AMHIP_ATAN_HD DD dd_mul(const DD& a, const DD& b) {
  DD p = dd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_fast_two_sum(p.hi, p.lo);
}
AMHIP_ATAN_HD DD dd_div(const DD& a, const DD& b) {
  const double q1 = a.hi / b.hi;
  // r = a - q1 * b
  DD p = dd_mul({q1, 0.0}, b);
  DD r = dd_add(a, {-p.hi, -p.lo});
  const double q2 = r.hi / b.hi;
  p = dd_mul({q2, 0.0}, b);
  r = dd_add(r, {-p.hi, -p.lo});
  const double q3 = r.hi / b.hi;
  DD q = dd_fast_two_sum(q1, q2);
  q.lo += q3;
  return dd_fast_two_sum(q.hi, q.lo);
}

AMHIP_ATAN_HD double atan_cr(double x) {
  constexpr double kTab[kAtanCrSteps + 1][2] = AMHIP_ATAN_CR_TABLE;
  constexpr double kCoeff[9][2] = AMHIP_ATAN_CR_COEFF;
  if (!(x == x)) return x;                       // NaN
  const double sign = x < 0.0 ? -1.0 : 1.0;      // (the callers pass radii; kept total: atan is odd)
  const double r = x < 0.0 ? -x : x;
  if (r < 0x1p-27) return x;                     // r - r^3 / 3 rounds to r
  if (r > 0x1p+60) return sign * kHalfPiHi;      // pi/2 - 1/r rounds to fl(pi/2)
  const bool invert = r > 1.0;
  DD t = {r, 0.0};
  if (invert) t = dd_div({1.0, 0.0}, {r, 0.0});
  const int j = (int)(t.hi * (double)kAtanCrSteps + 0.5);   // 0 .. 32
  const double c = (double)j * (1.0 / (double)kAtanCrSteps);
  DD y = t;
  if (j != 0) {
    const DD num = dd_add(t, {-c, 0.0});
    DD tc = dd_two_prod(t.hi, c);
    tc.lo += t.lo * c;
    const DD den = dd_add({1.0, 0.0}, dd_fast_two_sum(tc.hi, tc.lo));
    y = dd_div(num, den);
  }
  const DD y2 = dd_mul(y, y);
  DD q = {kCoeff[8][0], kCoeff[8][1]};
#pragma unroll
  for (int k = 7; k >= 0; --k) q = dd_add(dd_mul(q, y2), {kCoeff[k][0], kCoeff[k][1]});
  // atan y = y + y * (y2 * q)
  const DD corr = dd_mul(y, dd_mul(y2, q));
  DD a = dd_add(y, corr);
  a = dd_add({kTab[j][0], kTab[j][1]}, a);
  if (invert) a = dd_add({kHalfPiHi, kHalfPiLo}, {-a.hi, -a.lo});
  return sign * (a.hi + a.lo);
}

// The same value on a fast path (Ziv's strategy): atan as hi + lo with ~76 good bits from plain
// double arithmetic -- the quotient y and the cubic term with their rounding errors carried, the
// series' tail (y^5 / 5 ...: 2^-26 of the result) in double -- and a rounding test: when every
// value within 2^-70 (relative) of hi + lo rounds to the same double, that double is the correctly
// rounded atan (*ok = true); otherwise (one call in ~65 000) the caller takes atan_cr().
// r in [2^-27, 2^60] (outside: atan_cr's own shortcuts).
AMHIP_ATAN_HD double atan_cr_fast(double r, bool* ok) {
  constexpr double kTab[kAtanCrSteps + 1][2] = AMHIP_ATAN_CR_TABLE;
  constexpr double kCoeff[9][2] = AMHIP_ATAN_CR_COEFF;
  const bool invert = r > 1.0;
  double th = r, tl = 0.0;
  if (invert) {
    th = 1.0 / r;
    tl = fma(-th, r, 1.0) * th;  // 1 / r = th + tl
  }
  const int j = (int)(th * (double)kAtanCrSteps + 0.5);
  const double c = (double)j * (1.0 / (double)kAtanCrSteps);
  // y = (t - c) / (1 + t c), quotient and remainder
  const double n_hi = th - c;  // exact (Sterbenz; c = 0: th itself)
  const double p = th * c;
  const double pe = fma(th, c, -p) + tl * c;
  const double d_hi = 1.0 + p;
  const double d_lo = ((1.0 - d_hi) + p) + pe;
  const double y_hi = n_hi / d_hi;
  const double rem = fma(-y_hi, d_hi, n_hi) + (tl - y_hi * d_lo);
  const double y_lo = rem / d_hi;
  // atan y = y - y^3 / 3 + (y^5 / 5 - ... + y^13 / 13)
  const double s = y_hi * y_hi;
  const double s_lo = fma(y_hi, y_hi, -s) + 2.0 * y_hi * y_lo;
  double q = kCoeff[5][0];                 // 1/13
  q = fma(q, s, kCoeff[4][0]);             // -1/11
  q = fma(q, s, kCoeff[3][0]);             // 1/9
  q = fma(q, s, kCoeff[2][0]);             // -1/7
  q = fma(q, s, kCoeff[1][0]);             // 1/5
  const double tail = (y_hi * s) * (s * q);
  const double t3_hi = y_hi * s;
  const double t3_lo = fma(y_hi, s, -t3_hi) + (y_hi * s_lo + y_lo * s);
  const double m_hi = t3_hi * kCoeff[0][0];  // -1/3
  const double m_lo = fma(t3_hi, kCoeff[0][0], -m_hi) + (t3_hi * kCoeff[0][1] + t3_lo * kCoeff[0][0]);
  const double e = (m_lo + tail) + y_lo;  // (d/dy of the cubic at y_lo is inside t3_lo already)
  const DD a = dd_two_sum(kTab[j][0], y_hi);
  const DD b = dd_two_sum(a.hi, m_hi);
  DD res = dd_fast_two_sum(b.hi, ((a.lo + b.lo) + kTab[j][1]) + e);
  if (invert) {
    const DD h = dd_two_sum(kHalfPiHi, -res.hi);
    res = dd_fast_two_sum(h.hi, (h.lo + kHalfPiLo) - res.lo);
  }
  const double err = 0x1p-70 * res.hi;
  const double r1 = res.hi + (res.lo - err), r2 = res.hi + (res.lo + err);
  *ok = r1 == r2;
  return res.hi + res.lo;  // (== r1 == r2 when ok)
}

// What the device's distortion calls: the fast path's value WITHOUT the second step -- hi + lo
// carries ~76 good bits, so its rounding is the correctly rounded atan unless the exact value lies
// within 2^-76 (relative) of a rounding boundary: one call in ~8 million, against the one call
// in a thousand where the host's own libm is an ulp off.  No call, no double-double chain in the
// per-pair loop of k_ortho_backward (a not-inlined atan_cr() there cost 250 spilled registers
// and doubled the kernel's time for every distortion model).  Total: any r >= 0, inf, NaN.
AMHIP_ATAN_HD double atan_device(double r) {
  bool ok;  // (unused: see above)
  const double in_range = atan_cr_fast(fmin(fmax(r, 0x1p-27), 0x1p+60), &ok);
  return !(r >= 0x1p-27) ? r : (r > 0x1p+60 ? kHalfPiHi : in_range);  // (NaN, tiny: r itself)
}

}  // namespace amhip

#endif  // AMHIP_ATAN_CR_H_
