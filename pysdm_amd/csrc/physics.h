// physics.h -- per-pair / per-droplet device formulae shared by the fine-grained kernels and the
// fused step.  Evaluation order follows the reference's chain of Storage ops one by one (each op
// rounds once; the library is built with -ffp-contract=off so nothing is fused into an FMA).
#pragma once
#include "common.h"

#ifdef __HIPCC__

// np.sign(x) * np.power(np.abs(x), p)   (storage_impl.py:75-78)
__device__ __forceinline__ double signed_pow(double x, double p) {
  const double sg = (double)((x > 0) - (x < 0));
  return (x != x) ? x : sg * sdm_pow(fabs(x), p);
}

// `x **= 2` on a Storage: the sign survives (storage_impl.py:76-78), the square is exact
__device__ __forceinline__ double signed_sq(double x) {
  const double sg = (double)((x > 0) - (x < 0));
  return (x != x) ? x : sg * (x * x);
}

// attributes/physics/volume.py:16-17 + liquid_spheres.py:18-19
__device__ __forceinline__ double volume_of_mass(double m, double rho_w) { return m / rho_w; }

// attributes/physics/radius.py:15-17: product(volume, 1/PI_4_3); **= 1/3
__device__ __forceinline__ double radius_of_volume(double v, double inv_pi_4_3) {
  return signed_pow(v * inv_pi_4_3, 1.0 / 3.0);
}

// terminal_velocity_methods.py:14-25 (Gunn-Kinzer table: a = values, b = slopes)
__device__ __forceinline__ double gk_interpolate(double r, double factor,
                                                 const double *__restrict__ a,
                                                 const double *__restrict__ b,
                                                 int64_t table_len) {
  if (r < 0) return 0.0;
  const double x = factor * r;
  int64_t r_id = (int64_t)x;
  r_id = r_id > table_len - 1 ? table_len - 1 : r_id;  // memory safety (reference raises)
  const double r_rest = fmod(x, 1.0) / factor;
  return a[r_id] + r_rest * b[r_id];
}

// collisions_methods.py:743-769, one pair
__device__ __forceinline__ double linear_collection_efficiency(const double *__restrict__ P,
                                                               double ra, double rb,
                                                               double unit) {
  double r, r_s;
  if (ra > rb) { r = ra / unit; r_s = rb / unit; } else { r = rb / unit; r_s = ra / unit; }
  const double p = r_s / r;
  double out = 0.0;
  if (p != 0 && p != 1) {
    const double G = sdm_pow(P[8] / r, P[12]) + P[9] + P[10] * r;
    const double Gp = sdm_pow(1 - p, G);
    if (Gp != 0) {
      const double D = P[2] / sdm_pow(r, P[3]);
      const double E = P[4] / sdm_pow(r, P[5]);
      const double F = sdm_pow(P[6] / r, P[11]) + P[7];
      const double v = P[0] + P[1] * p + D / sdm_pow(p, F) + E / Gp;
      out = v > 0 ? v : 0;
    }
  }
  return out;
}

// fragmentation_methods.py:76-95, one pair (nfmax < 0 == None)
__device__ __forceinline__ void fragmentation_limiters(double &n_fragment, double &frag_volume,
                                                       double vmin, double nfmax,
                                                       double x_plus_y) {
  if (x_plus_y == 0.0) {
    frag_volume = 0.0;
    n_fragment = 1.0;
  } else {
    if (frag_volume != frag_volume || frag_volume == 0.0) frag_volume = x_plus_y;
    frag_volume = frag_volume < x_plus_y ? frag_volume : x_plus_y;
    if (nfmax >= 0 && x_plus_y / frag_volume > nfmax)
      frag_volume = x_plus_y / nfmax;
    else if (frag_volume < vmin)
      frag_volume = x_plus_y;
    n_fragment = x_plus_y / frag_volume;
  }
}

// physics/trivia.py:95-108
__device__ __forceinline__ double erfinv_approx(double c, double VA, double Vb) {
  return 2 * sqrt(VA) * sdm_sinh(sdm_asinh(sdm_atanh(c) / 2 / Vb / sdm_pow(VA, 1.5)) / 3);
}


/* Python's max(a, b) = b if b > a else a ; min(a, b) = b if b < a else a */
#define PYMAX(a, b) ((b) > (a) ? (b) : (a))
#define PYMIN(a, b) ((b) < (a) ? (b) : (a))
#define LL_PI 3.141592653589793

/* Low & List 1982 fragment-size parameters, PySDM/physics/fragmentation_function/lowlist82.py
 * (each returns H, mu, sigma in cm units; the fixed-point loops run 10 rounds as there) */
struct LL82P { double H, mu, sigma; };

__device__ inline struct LL82P ll82_gauss_fixed_point(double H, double mu, double upper) {
  /* :22-30, :108-116, :150-158: sigma <- sqrt(2/pi)/H / (1 + sdm_erf((upper - mu)/(sqrt2 sigma))) */
  double sigma = 1 / H;
  for (int r = 0; r < 10; ++r)
    sigma = 1 / H * sqrt(2 / LL_PI) / (1 + sdm_erf((upper - mu) / (sqrt(2.0) * sigma)));
  struct LL82P p = {H, mu, sigma};
  return p;
}

__device__ inline struct LL82P ll82_f1(double CM, double dl, double dcoal) { /* :15-30 */
  const double dlCM = dl / CM;
  return ll82_gauss_fixed_point(50.8 * sdm_pow(dlCM, -0.718), dlCM, dcoal / CM);
}

__device__ inline struct LL82P ll82_f2(double CM, double ds) { /* :33-38 */
  const double dsCM = ds / CM;
  const double H = 4.18 * sdm_pow(dsCM, -1.17);
  struct LL82P p = {H, dsCM, 1 / (sqrt(2 * LL_PI) * H)};
  return p;
}

__device__ inline struct LL82P ll82_f3(double CM, double ds, double dl) { /* :41-98 */
  const double dsCM = ds / CM, dlCM = dl / CM;
  double Ff1 = (-2.25e4 * sdm_pow(dlCM - 0.403, 2.0) - 37.9) * sdm_pow(dsCM, 2.5) +
               9.67 * sdm_pow(dlCM - 0.170, 2.0) + 4.95;
  Ff1 = PYMAX(0.0, Ff1);
  const double Ff2 = 1.02e4 * sdm_pow(dsCM, 2.83) + 2;
  const double ds0 = PYMAX(0.04, sdm_pow(Ff1 / 2.83, 1 / 1.02e4));
  const double Ff = dsCM > ds0 ? PYMAX(2.0, Ff1) : PYMAX(2.0, Ff2);
  const double Dff3 = 0.241 * dsCM + 0.0129;
  const double Pf301 = 1.68e5 * sdm_pow(dsCM, 2.33);
  const double Pf302 = PYMAX(0.0, (43.4 * sdm_pow(dlCM + 1.81, 2.0) - 159.0) / dsCM -
                                      3870 * sdm_pow(dlCM - 0.285, 2.0) - 58.1);
  const double alpha = (dsCM - ds0) / (0.2 * ds0);
  const double Pf303 = alpha * Pf301 + (1 - alpha) * Pf302;
  const double Pf0 = dsCM < ds0 ? Pf301 : (dsCM > 1.2 * ds0 ? Pf302 : Pf303);
  double sigma = 10 * Dff3;
  double mu = sdm_log(Dff3) + sigma * sigma;
  double H = Pf0 * Dff3 / sdm_exp(-0.5 * (sigma * sigma));
  for (int r = 0; r < 10; ++r) {
    if (sigma == 0.0 || H == 0) {
      struct LL82P z = {0.0, sdm_log(ds0), sdm_log(ds0)};
      return z;
    }
    sigma = sqrt(2 / LL_PI) * (Ff - 2) / H / (1 - sdm_erf((sdm_log(0.01) - mu) / sqrt(2.0) / sigma));
    mu = sdm_log(Dff3) + sigma * sigma;
    H = Pf0 * Dff3 / sdm_exp(-0.5 * (sigma * sigma));
  }
  struct LL82P p = {H, mu, sigma};
  return p;
}

__device__ inline struct LL82P ll82_s1(double CM, double dl, double ds, double dcoal) { /* :100-116 */
  return ll82_gauss_fixed_point(100 * sdm_exp(-3.25 * (ds / CM)), dl / CM, dcoal / CM);
}

__device__ inline struct LL82P ll82_s2(double CM, double dl, double ds, double St) { /* :118-143 */
  const double dsCM = ds / CM, dlCM = dl / CM;
  const double Dss2 = 0.254 * sdm_pow(dsCM, 0.413) * sdm_exp(3.53 * sdm_pow(dsCM, 2.51) * (dlCM - dsCM));
  const double bstar = 14.2 * sdm_exp(-17.2 * dsCM);
  const double Ps20 = 0.23 * sdm_pow(dsCM, -3.93) * sdm_pow(dlCM, bstar);
  double sigma = 10 * Dss2;
  double mu = sdm_log(Dss2) + sigma * sigma;
  double H = Ps20 * Dss2 / sdm_exp(-0.5 * (sigma * sigma));
  const double Fs = 5 * sdm_erf((St - 2.52e-6) / (1.85e-6)) + 6;
  for (int r = 0; r < 10; ++r) {
    sigma = sqrt(2 / LL_PI) * (Fs - 1) / H / (1 - sdm_erf((sdm_log(0.01) - mu) / sqrt(2.0) / sigma));
    mu = sdm_log(Dss2) + sigma * sigma;
    H = Ps20 * Dss2 / sdm_exp(-0.5 * (sigma * sigma));
  }
  struct LL82P p = {H, mu, sigma};
  return p;
}

__device__ inline struct LL82P ll82_d1(double CM, double W1, double dl, double dcoal, double CKE) {
  /* :145-160 */
  const double mu = (dl / CM) * (1 - sdm_exp(-3.70 * (3.10 - W1)));
  return ll82_gauss_fixed_point(1.58e-5 * sdm_pow(CKE, -1.22), mu, dcoal / CM);
}

__device__ inline struct LL82P ll82_d2(double CM, double ds, double dl, double CKE) { /* :162-193 */
  const double dsCM = ds / CM, dlCM = dl / CM;
  const double Ddd2 = sdm_exp(-17.4 * dsCM - 0.671 * (dlCM - dsCM)) * dsCM;
  const double bstar = 0.007 * sdm_pow(dsCM, -2.54);
  const double Pd20 = 0.0884 * sdm_pow(dsCM, -2.52) * sdm_pow(dlCM - dsCM, bstar);
  double sigma = 10 * Ddd2;
  double mu = sdm_log(Ddd2) + sigma * sigma;
  double H = Pd20 * Ddd2 / sdm_exp(-0.5 * (sigma * sigma));
  const double Fd = PYMAX(1.0, 297.5 + 23.7 * sdm_log(CKE));
  struct LL82P z = {0.0, sdm_log(Ddd2), sdm_log(Ddd2)};
  if (Fd == 1.0) return z;
  for (int r = 0; r < 10; ++r) {
    if (sigma == 0.0 || H <= 0.1) return z;
    if (sigma >= 1.0) return z;
    sigma = sqrt(2 / LL_PI) * (Fd - 1) / H / (1 - sdm_erf((sdm_log(0.01) - mu) / sqrt(2.0) / sigma));
    mu = sdm_log(Ddd2) + sigma * sigma;
    H = Pd20 * Ddd2 / sdm_exp(-0.5 * (sigma * sigma));
  }
  struct LL82P p = {H, mu, sigma};
  return p;
}

/* one pair of fragmentation_methods.py:379-474 (+ ll82_Nr :51-72); *rand, *Rf, *Rs, *Rd are
 * updated in place as the reference does; K = {CM, PI, VEDDER_1987_A, VEDDER_1987_b} */
__device__ inline double ll82_fragment_volume(double CKE, double W, double W2, double St, double ds, double dl,
                                 double dcoal, double *rand, double *Rf, double *Rs, double *Rd,
                                 double tol, const double *K) {
  const double CM = K[0], PI = K[1], VA = K[2], Vb = K[3];
  if (dl <= 0.4e-3) return sdm_pow(dcoal, 3.0) * PI / 6;
  if (ds == 0.0 || dl == 0.0) return 1e-18;
  *Rf = CKE >= 0.893e-6 ? 1.11e-4 * sdm_pow(CKE, -0.654) : 1.0;
  *Rs = W >= 0.86 ? 0.685 * (1 - sdm_exp(-1.63 * (W2 - 0.86))) : 0.0;
  *Rd = (*Rs + *Rf) > 1.0 ? 0.0 : 1.0 - *Rs - *Rf;
  double d;  /* fragment diameter in cm */
  if (*rand <= *Rf) {  /* filament breakup */
    const struct LL82P p1 = ll82_f1(CM, dl, dcoal), p2 = ll82_f2(CM, ds), p3 = ll82_f3(CM, ds, dl);
    const double H1 = p1.H * p1.mu, H2 = p2.H * p2.mu, H3 = p3.H * sdm_exp(p3.mu);
    const double Hsum = H1 + H2 + H3;
    *rand = *rand / *Rf;
    if (*rand <= H1 / Hsum) {
      const double X = PYMAX(*rand * Hsum / H1, tol);
      d = p1.mu + sqrt(2.0) * p1.sigma * erfinv_approx(2 * X - 1, VA, Vb);
    } else if (*rand <= (H1 + H2) / Hsum) {
      const double X = (*rand * Hsum - H1) / H2;
      d = p2.mu + sqrt(2.0) * p2.sigma * erfinv_approx(2 * X - 1, VA, Vb);
    } else {
      const double X = PYMIN((*rand * Hsum - H1 - H2) / H3, 1.0 - tol);
      d = sdm_exp(p3.mu + sqrt(2.0) * p3.sigma * erfinv_approx(2 * X - 1, VA, Vb));
    }
  } else if (*rand <= *Rf + *Rs) {  /* sheet breakup */
    const struct LL82P p1 = ll82_s1(CM, dl, ds, dcoal), p2 = ll82_s2(CM, dl, ds, St);
    const double H1 = p1.H * p1.mu, H2 = p2.H * sdm_exp(p2.mu);
    const double Hsum = H1 + H2;
    *rand = (*rand - *Rf) / (*Rs);
    if (*rand <= H1 / Hsum) {
      const double X = PYMAX(*rand * Hsum / H1, tol);
      d = p1.mu + sqrt(2.0) * p1.sigma * erfinv_approx(2 * X - 1, VA, Vb);
    } else {
      const double X = PYMIN((*rand * Hsum - H1) / H2, 1.0 - tol);
      d = sdm_exp(p2.mu + sqrt(2.0) * p2.sigma * erfinv_approx(2 * X - 1, VA, Vb));
    }
  } else {  /* disk breakup */
    const struct LL82P p1 = ll82_d1(CM, W, dl, dcoal, CKE), p2 = ll82_d2(CM, ds, dl, CKE);
    const double H1 = p1.H * p1.mu, H2 = p2.H;
    const double Hsum = H1 + H2;
    *rand = (*rand - *Rf - *Rs) / *Rd;
    if (*rand <= H1 / Hsum) {
      const double X = PYMAX(*rand * Hsum / H1, tol);
      d = p1.mu + sqrt(2.0) * p1.sigma * erfinv_approx(2 * X - 1, VA, Vb);
    } else {
      const double X = PYMIN((*rand * Hsum - H1) / H2, 1 - tol);
      d = sdm_exp(p2.mu + sqrt(2.0) * p2.sigma * erfinv_approx(2 * X - 1, VA, Vb));
    }
  }
  d = d * 0.01;  /* cm -> m */
  return sdm_pow(d, 3.0) * PI / 6;
}

struct StraubTmp { double Nr1, Nr2, Nr3, Nr4, Nrt, d34; };

// fragmentation_methods.py:321-377 (+ :12-48, physics/fragmentation_function/straub2010nf.py)
// T holds the incoming Nr* values (the reference zero-fills them before the call).
__device__ __forceinline__ double straub_fragment_volume(double CW, double gam, double ds,
                                                         double v_max, double rand,
                                                         const double *__restrict__ K,
                                                         StraubTmp &T) {
  const double CM = K[0], E_D1 = K[1], MU2 = K[2], VA = K[3], Vb = K[4], PI = K[5];
  if (gam * CW >= 7.0) T.Nr1 = 0.088 * (gam * CW - 7.0);
  if (CW >= 21.0) {
    T.Nr2 = 0.22 * (CW - 21.0);
    if (CW <= 46.0) T.Nr3 = 0.04 * (46.0 - CW);
  } else {
    T.Nr3 = 1.0;
  }
  T.Nr4 = 1.0;
  T.Nrt = T.Nr1 + T.Nr2 + T.Nr3 + T.Nr4;
  const double sigma1 = sqrt(sdm_log(CW / 64 / 100 * CM * CM / 12 / sdm_pow(E_D1, 2.0) + 1));
  const double mu1 = sdm_log(E_D1) - sdm_pow(sigma1, 2.0) / 2;
  const double s2a = 7 * (CW - 21) * CM / 1000;
  const double sigma2 = (s2a > 0.0 ? s2a : 0.0) / sqrt(12.0);
  const double mu2 = MU2;
  const double sigma3 = (1 + 0.76 * sqrt(CW)) * CM / 100 / sqrt(12.0);
  const double mu3 = 0.9 * ds;
  T.Nr1 = T.Nr1 * sdm_exp(3 * mu1 + 9 * sdm_pow(sigma1, 2.0) / 2);
  T.Nr2 = T.Nr2 * (sdm_pow(mu2, 3.0) + 3 * mu2 * sdm_pow(sigma2, 2.0));
  T.Nr3 = T.Nr3 * (sdm_pow(mu3, 3.0) + 3 * mu3 * sdm_pow(sigma3, 2.0));
  T.Nr4 = v_max * 6 / PI + sdm_pow(ds, 3.0) - T.Nr1 - T.Nr2 - T.Nr3;
  if (T.Nr4 <= 0.0) {
    T.d34 = 0;
    T.Nr4 = 0;
  } else {
    T.d34 = sdm_exp(sdm_log(T.Nr4) / 3);
  }
  T.Nrt = T.Nr1 + T.Nr2 + T.Nr3 + T.Nr4;
  double diameter;
  if (T.Nrt == 0.0) {
    diameter = 0.0;
  } else if (rand < T.Nr1 / T.Nrt) {
    const double X = rand * T.Nrt / T.Nr1;
    diameter = sdm_exp(mu1 + sqrt(2.0) * sigma1 * erfinv_approx(X, VA, Vb));
  } else if (rand < (T.Nr2 + T.Nr1) / T.Nrt) {
    const double X = (rand * T.Nrt - T.Nr1) / T.Nr2;
    diameter = mu2 + sqrt(2.0) * sigma2 * erfinv_approx(X, VA, Vb);
  } else if (rand < (T.Nr3 + T.Nr2 + T.Nr1) / T.Nrt) {
    const double X = (rand * T.Nrt - T.Nr1 - T.Nr2) / T.Nr3;
    diameter = mu3 + sqrt(2.0) * sigma3 * erfinv_approx(X, VA, Vb);
  } else {
    diameter = T.d34;
  }
  return sdm_pow(diameter, 3.0) * PI / 6;
}

// ---- multiplicity / attribute update, collisions_methods.py:44-243 ----------------------
// coalesce :44-59 (counter add done by the caller)
__device__ __forceinline__ void coalesce_pair(int64_t j, int64_t k, double gamma,
                                              int64_t *__restrict__ multiplicity,
                                              double *__restrict__ attributes, int64_t n_attr,
                                              int64_t n_sd) {
  const int64_t nj = multiplicity[j], nk = multiplicity[k];
  const double new_n = (double)nj - gamma * (double)nk;
  if (new_n > 0) {
    multiplicity[j] = (int64_t)new_n;
    for (int64_t a = 0; a < n_attr; ++a)
      attributes[a * n_sd + k] += gamma * attributes[a * n_sd + j];
  } else {
    const int64_t half = nk / 2;
    multiplicity[j] = half;
    multiplicity[k] = nk - half;
    for (int64_t a = 0; a < n_attr; ++a) {
      const double v = gamma * attributes[a * n_sd + j] + attributes[a * n_sd + k];
      attributes[a * n_sd + j] = v;
      attributes[a * n_sd + k] = v;
    }
  }
}

// :62-93.  The reference's loop runs `int(gamma)` successive breakups of one pair, each depending
// on the last in floating point (two roundings per iteration: there is no closed form), and late in
// a breakup run gamma reaches 1e4..1e6 for single pairs - one lane walks the loop alone while its
// kernel waits.  What can be saved is everything but the dependent arithmetic: the loop below
// advances 8 iterations at a time without a branch (the values of an iteration that turns out to
// be beyond the exit are simply not used) and tests the two exit conditions of all 8 at once;
// the block in which an exit falls is walked again one iteration at a time.  Same operations, same
// order, same results as the one-at-a-time loop (tests/micro_cases.py: the reference's breakup
// answers; tests/test_hip_parity.py: long-gamma cases against the checker).
__device__ __forceinline__ void compute_transfer_multiplicities(
    double gamma, int64_t nj, int64_t nk, double mj, double mk, double fragment_mass_i,
    int64_t max_multiplicity, double &take_from_j, double &new_mult_k, int64_t &gamma_j_k,
    bool &overflow) {
  overflow = false;
  gamma_j_k = 0;
  double t = (double)nk;  // take_from_j_test
  take_from_j = 0;
  double x = ((mj + mk) / fragment_mass_i) * (double)nk;  // new_mult_k_test
  new_mult_k = (double)nk;
  const int64_t g = (int64_t)gamma;
  const double r = mj / fragment_mass_i, top = (double)max_multiplicity, have = (double)nj;
  int64_t m = 0;
#ifdef SDM_BREAKUP_ONE_AT_A_TIME  // (A/B measurement build: profiles/README.md)
  constexpr int B = 1 << 30;
#else
  constexpr int B = 8;
#endif
  while (m + B <= g) {
    double xs[B + 1], ts[B + 1];
    xs[0] = x;
    ts[0] = t;
    bool exit_inside = false;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      exit_inside |= (xs[b] > top) | (ts[b] > have);
      ts[b + 1] = ts[b] + xs[b];
      xs[b + 1] = xs[b] * r + xs[b];
    }
    if (exit_inside) break;  // some iteration of this block leaves the loop: one at a time below
    take_from_j = ts[B - 1];
    new_mult_k = xs[B - 1];
    m += B;
    x = xs[B];
    t = ts[B];
  }
  gamma_j_k = m;
  for (; m < g; ++m) {
    const bool over = x > top;
    if (over | (t > have)) { overflow = over; break; }
    take_from_j = t;
    new_mult_k = x;
    gamma_j_k = m + 1;
    t += x;
    x = x * r + x;
  }
}

// :96-132 fused: new multiplicities, attribute transfer, rounding with attribute rescale
__device__ __forceinline__ void apply_breakup_transfer(int64_t j, int64_t k, double take_from_j,
                                                       double new_mult_k,
                                                       int64_t *__restrict__ multiplicity,
                                                       double *__restrict__ attributes,
                                                       int64_t n_attr, int64_t n_sd) {
  const int64_t nj0 = multiplicity[j], nk0 = multiplicity[k];
  double nj, nk;
  const bool split = !((double)nj0 > take_from_j);
  if (!split) { nj = (double)nj0 - take_from_j; nk = new_mult_k; }
  else { nj = new_mult_k / 2; nk = nj; }
  const int64_t rj = py_round(nj), rk = py_round(nk);
  const int64_t ij = rj > 1 ? rj : 1, ik = rk > 1 ? rk : 1;
  const double factor_j = nj / (double)ij, factor_k = nk / (double)ik;
  for (int64_t a = 0; a < n_attr; ++a) {
    double ak = attributes[a * n_sd + k], aj = attributes[a * n_sd + j];
    ak *= (double)nk0;
    ak += take_from_j * aj;
    ak /= new_mult_k;
    if (split) aj = ak;
    ak *= factor_k;
    aj *= factor_j;
    attributes[a * n_sd + k] = ak;
    attributes[a * n_sd + j] = aj;
  }
  multiplicity[j] = ij;
  multiplicity[k] = ik;
}

#endif  // __HIPCC__
