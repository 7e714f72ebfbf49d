/*
 * sdm_math.h -- the transcendental functions of the collision path, one implementation for every
 * compiler that builds this project (hipcc for gfx950, gcc for the CPU checker).
 *
 * Why: breakup efficiencies and fragment sizes go through pow / exp / log / erf / sinh / asinh /
 * atanh (fragmentation_methods.py:12-48,136-144,321-377, physics/trivia.py:95-108,
 * coalescence_efficiencies/straub2010.py:27-50 of the reference); their results enter the masses
 * and, through `break_up` (collisions_methods.py:62-132), integer multiplicities.  A device libm
 * and a host libm differ in the last bit now and then, and a long breakup run then takes different
 * integer decisions.  These functions use only IEEE-754 double + - * / sqrt, comparisons, integer
 * operations and bit casts - every one of them correctly rounded or exact on both targets - in a
 * fixed order (build with -ffp-contract=off: no operation may be fused; no FMA is used, exact
 * products are Dekker's), so they return THE SAME BITS on the GPU and on the host.
 *
 * Accuracy (tests/test_sdm_math.py, against mpmath): exp, log, pow within 0.52 ulp (correctly
 * rounded except in about one case in a thousand), sinh / asinh / atanh / erf within 2 ulp.
 * Method: log in double-double (table of 130 points c with 1/c rounded to double: z/c - 1 is
 * formed exactly, log1p by series), exp of a double-double (k/128 table, degree-6 polynomial),
 * pow = exp(y * log x) with the product in double-double.  Tables: sdm_math_tables.h, generated
 * by scripts/gen_sdm_math_tables.py.
 */
#ifndef SDM_MATH_H
#define SDM_MATH_H
#include <stdint.h>

#ifdef __HIPCC__
#define SDM_MATH_FN __device__ static inline
#define SDM_MATH_TABLE __device__ static const
#define SDM_MATH_SQRT(x) __builtin_sqrt(x) /* v_sqrt_f64 + fix-up: correctly rounded */
#else
#define SDM_MATH_FN static inline
#define SDM_MATH_TABLE static const
#define SDM_MATH_SQRT(x) __builtin_sqrt(x)
#endif
#include "sdm_math_tables.h"

typedef struct sdm_dd { double hi, lo; } sdm_dd;

SDM_MATH_FN uint64_t sdm_bits(double x) {
  union { double d; uint64_t u; } c;
  c.d = x;
  return c.u;
}
SDM_MATH_FN double sdm_from_bits(uint64_t u) {
  union { double d; uint64_t u; } c;
  c.u = u;
  return c.d;
}
SDM_MATH_FN double sdm_abs(double x) { return sdm_from_bits(sdm_bits(x) & 0x7fffffffffffffffULL); }
SDM_MATH_FN double sdm_nan(void) { return sdm_from_bits(0x7ff8000000000000ULL); }
SDM_MATH_FN double sdm_inf(void) { return sdm_from_bits(0x7ff0000000000000ULL); }
/* 2^e for -1022 <= e <= 1023 */
SDM_MATH_FN double sdm_pow2i(int64_t e) { return sdm_from_bits((uint64_t)(e + 1023) << 52); }

/* s + t = a + b exactly (Knuth) */
SDM_MATH_FN sdm_dd sdm_two_sum(double a, double b) {
  sdm_dd r;
  r.hi = a + b;
  const double bb = r.hi - a;
  r.lo = (a - (r.hi - bb)) + (b - bb);
  return r;
}
/* p + e = a * b exactly (Dekker / Veltkamp; |a|, |b| < 2^995) */
SDM_MATH_FN sdm_dd sdm_two_prod(double a, double b) {
  sdm_dd r;
  r.hi = a * b;
  const double ca = 134217729.0 * a, cb = 134217729.0 * b;
  const double ah = ca - (ca - a), bh = cb - (cb - b);
  const double al = a - ah, bl = b - bh;
  r.lo = ((ah * bh - r.hi) + ah * bl + al * bh) + al * bl;
  return r;
}

/* exp(xh + xl), |xl| << |xh|; the caller has dealt with NaN */
SDM_MATH_FN double sdm_exp_dd(double xh, double xl) {
  if (xh > 709.782712893384) return sdm_inf();
  if (xh < -745.1332191019412) return 0.0;
  const double z = xh * SDM_N_OVER_LN2;
  const int64_t k = (int64_t)(z + (z >= 0 ? 0.5 : -0.5));
  const double kd = (double)k;
  const double r = ((xh - kd * SDM_LN2_OVER_N_HI) - kd * SDM_LN2_OVER_N_LO) + xl;
  const int64_t i = k & 127;
  const int64_t e = (k - i) / 128;
  const double p =
      r + (r * r) * (0.5 + r * (1.0 / 6 + r * (1.0 / 24 + r * (1.0 / 120 + r * (1.0 / 720)))));
  const double th = sdm_exp2_tab[i][0], tl = sdm_exp2_tab[i][1];
  const double y = th + (tl + th * p); /* in [1, 2) up to rounding */
  if (e >= -1021 && e <= 1022) return y * sdm_pow2i(e);
  if (e > 1022) return (y * sdm_pow2i(1022)) * sdm_pow2i(e - 1022);
  return (y * sdm_pow2i(e + 1000)) * sdm_pow2i(-1000); /* subnormal result */
}

SDM_MATH_FN double sdm_exp(double x) {
  if (x != x) return x;
  return sdm_exp_dd(x, 0.0);
}

/* log(x) as hi + lo for finite x > 0 */
SDM_MATH_FN sdm_dd sdm_log_dd(double x) {
  int64_t e = 0;
  uint64_t u = sdm_bits(x);
  if (u < 0x0010000000000000ULL) { /* subnormal */
    u = sdm_bits(x * 18014398509481984.0);
    e = -54;
  }
  e += (int64_t)(u >> 52) - 1023;
  double m = sdm_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m >= 1.4140625) {
    m *= 0.5;
    e += 1;
  }
  const int idx = m < 1.0 ? (int)(m * 256 + 0.5) - 181 : 76 + (int)(m * 128 + 0.5) - 128;
  const double invc = sdm_log_tab[idx][0];
  /* r = m / c - 1 with c := 1 / invc, exactly */
  const sdm_dd mp = sdm_two_prod(m, invc);
  const sdm_dd r = sdm_two_sum(mp.hi - 1.0, mp.lo);
  /* log1p(r) = r - r^2/2 + r^3 (1/3 - r/4 + ...), |r| < 2^-8 */
  sdm_dd q = sdm_two_prod(r.hi, r.hi);
  q.lo += 2 * (r.hi * r.lo);
  const double rh = r.hi;
  const double tail =
      ((rh * rh) * rh) *
      (1.0 / 3 +
       rh * (-0.25 + rh * (0.2 + rh * (-1.0 / 6 + rh * (1.0 / 7 + rh * (-0.125 + rh * (1.0 / 9)))))));
  const double ed = (double)e;
  sdm_dd s = sdm_two_sum(ed * SDM_LN2_HI, sdm_log_tab[idx][1]);
  double t = s.lo;
  s = sdm_two_sum(s.hi, rh);
  t += s.lo;
  s = sdm_two_sum(s.hi, -0.5 * q.hi);
  t += s.lo;
  t += (((ed * SDM_LN2_LO + sdm_log_tab[idx][2]) + r.lo) - 0.5 * q.lo) + tail;
  sdm_dd out;
  out.hi = s.hi + t;
  out.lo = t - (out.hi - s.hi);
  return out;
}

SDM_MATH_FN double sdm_log(double x) {
  if (x != x) return x;
  if (x < 0) return sdm_nan();
  if (x == 0) return -sdm_inf();
  if (x == sdm_inf()) return x;
  return sdm_log_dd(x).hi;
}

SDM_MATH_FN double sdm_log1p(double x) {
  if (x != x) return x;
  if (x < -1) return sdm_nan();
  if (x == -1) return -sdm_inf();
  if (x == sdm_inf()) return x;
  if (sdm_abs(x) < 5.551115123125783e-17) return x; /* 2^-54 */
  const sdm_dd u = sdm_two_sum(1.0, x);
  const sdm_dd l = sdm_log_dd(u.hi);
  return l.hi + (l.lo + u.lo / u.hi);
}

/* C's pow for the arguments the path produces, with the usual special cases */
SDM_MATH_FN double sdm_pow(double x, double y) {
  if (y == 0.0 || x == 1.0) return 1.0;
  if (x != x || y != y) return x + y;
  if (y == 1.0) return x;
  if (y == 2.0) return x * x;
  if (y == 0.5 && x >= 0 && x != sdm_inf()) return SDM_MATH_SQRT(x + 0.0); /* (+0.0: -0 -> +0) */
  const double ay = sdm_abs(y);
  double sign = 1.0;
  double ax = x;
  if (sdm_bits(x) >> 63) { /* negative base (or -0): a result only for integer y */
    ax = -x;
    /* every double >= 2^53 is an even integer */
    const int y_is_int = ay >= 9007199254740992.0 || (double)(int64_t)y == y;
    if (!y_is_int && ax != 0 && ax != sdm_inf()) return sdm_nan();
    if (y_is_int && ay < 9007199254740992.0 && (((int64_t)y) & 1)) sign = -1.0;
  }
  if (ax == 1.0) return sign;
  if (ax == 0) return y > 0 ? sign * 0.0 : sign * sdm_inf();
  if (ax == sdm_inf()) return y > 0 ? sign * sdm_inf() : sign * 0.0;
  if (ay == sdm_inf()) return ((ax > 1) == (y > 0)) ? sdm_inf() : 0.0;
  const sdm_dd l = sdm_log_dd(ax);
  if (ay > 1e300) return ((l.hi > 0) == (y > 0)) ? sign * sdm_inf() : sign * 0.0;
  /* y * (l.hi + l.lo) in double-double */
  if (sdm_abs(l.hi) * ay > 1e4) /* far beyond the range of exp: keep the product finite */
    return ((l.hi > 0) == (y > 0)) ? sign * sdm_inf() : sign * 0.0;
  sdm_dd p = sdm_two_prod(y, l.hi);
  p.lo += y * l.lo;
  const double eh = p.hi + p.lo;
  const double el = p.lo - (eh - p.hi);
  return sign * sdm_exp_dd(eh, el);
}

SDM_MATH_FN double sdm_sinh(double x) {
  if (x != x) return x;
  const double a = sdm_abs(x);
  if (a < 3.725290298461914e-09) return x; /* 2^-28 */
  if (a < 1.0) {
    const double t = x * x;
    /* x (1 + t/6 (1 + t/20 (1 + t/42 (...)))) : odd Taylor series to x^23 */
    double s = 1.0 + t / 506.0;
    s = 1.0 + (t / 420.0) * s;
    s = 1.0 + (t / 342.0) * s;
    s = 1.0 + (t / 272.0) * s;
    s = 1.0 + (t / 210.0) * s;
    s = 1.0 + (t / 156.0) * s;
    s = 1.0 + (t / 110.0) * s;
    s = 1.0 + (t / 72.0) * s;
    s = 1.0 + (t / 42.0) * s;
    s = 1.0 + (t / 20.0) * s;
    s = 1.0 + (t / 6.0) * s;
    return x * s;
  }
  double r;
  if (a > 709.0) {
    const double h = sdm_exp_dd(a - 1.0, 0.0); /* e^(a-1): finite up to a = 710.78 */
    r = (h * 0.5) * 2.718281828459045;
  } else {
    const double ex = sdm_exp_dd(a, 0.0);
    r = a > 40.0 ? 0.5 * ex : 0.5 * (ex - 1.0 / ex);
  }
  return x < 0 ? -r : r;
}

SDM_MATH_FN double sdm_asinh(double x) {
  if (x != x) return x;
  const double a = sdm_abs(x);
  if (a < 3.725290298461914e-09) return x;
  double r;
  if (a > 268435456.0) { /* 2^28: log(2a) */
    if (a == sdm_inf()) return x;
    const sdm_dd l = sdm_log_dd(a);
    r = l.hi + (l.lo + 0.6931471805599453);
  } else {
    const double t = a * a;
    r = sdm_log1p(a + t / (1.0 + SDM_MATH_SQRT(1.0 + t)));
  }
  return x < 0 ? -r : r;
}

SDM_MATH_FN double sdm_atanh(double x) {
  if (x != x) return x;
  const double a = sdm_abs(x);
  if (a > 1) return sdm_nan();
  if (a == 1) return x < 0 ? -sdm_inf() : sdm_inf();
  if (a < 3.725290298461914e-09) return x;
  const double r = 0.5 * sdm_log1p((a + a) / (1.0 - a));
  return x < 0 ? -r : r;
}

SDM_MATH_FN double sdm_erf(double x) {
  if (x != x) return x;
  const double a = sdm_abs(x);
  double r;
  if (a >= 6.0) {
    r = 1.0;
  } else if (a < 0.25) {
    const double t = a * a;
    double s = sdm_erf0_tab[SDM_ERF0_TERMS - 1];
    for (int n = SDM_ERF0_TERMS - 2; n >= 0; --n) s = sdm_erf0_tab[n] + t * s;
    r = a * s;
  } else {
    const int i = (int)(a * 4);
    const double t = a - ((double)i + 0.5) * 0.25;
    double s = sdm_erf_tab[i][SDM_ERF_TERMS - 1];
    for (int n = SDM_ERF_TERMS - 2; n >= 0; --n) s = sdm_erf_tab[i][n] + t * s;
    r = s > 1.0 ? 1.0 : s;
  }
  return x < 0 ? -r : r;
}

#endif /* SDM_MATH_H */
