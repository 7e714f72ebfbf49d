// physics.h -- per-pair / per-droplet device formulae shared by the fine-grained kernels and the
// fused step.  Evaluation order follows the reference's chain of Storage ops one by one (each op
// rounds once; the library is built with -ffp-contract=off so nothing is fused into an FMA).
#pragma once
#include "common.h"

#ifdef __HIPCC__

// np.sign(x) * np.power(np.abs(x), p)   (storage_impl.py:75-78)
__device__ __forceinline__ double signed_pow(double x, double p) {
  const double sg = (double)((x > 0) - (x < 0));
  return (x != x) ? x : sg * pow(fabs(x), p);
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
    const double G = pow(P[8] / r, P[12]) + P[9] + P[10] * r;
    const double Gp = pow(1 - p, G);
    if (Gp != 0) {
      const double D = P[2] / pow(r, P[3]);
      const double E = P[4] / pow(r, P[5]);
      const double F = pow(P[6] / r, P[11]) + P[7];
      const double v = P[0] + P[1] * p + D / pow(p, F) + E / Gp;
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
  return 2 * sqrt(VA) * sinh(asinh(atanh(c) / 2 / Vb / pow(VA, 1.5)) / 3);
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
  const double sigma1 = sqrt(log(CW / 64 / 100 * CM * CM / 12 / pow(E_D1, 2.0) + 1));
  const double mu1 = log(E_D1) - pow(sigma1, 2.0) / 2;
  const double s2a = 7 * (CW - 21) * CM / 1000;
  const double sigma2 = (s2a > 0.0 ? s2a : 0.0) / sqrt(12.0);
  const double mu2 = MU2;
  const double sigma3 = (1 + 0.76 * sqrt(CW)) * CM / 100 / sqrt(12.0);
  const double mu3 = 0.9 * ds;
  T.Nr1 = T.Nr1 * exp(3 * mu1 + 9 * pow(sigma1, 2.0) / 2);
  T.Nr2 = T.Nr2 * (pow(mu2, 3.0) + 3 * mu2 * pow(sigma2, 2.0));
  T.Nr3 = T.Nr3 * (pow(mu3, 3.0) + 3 * mu3 * pow(sigma3, 2.0));
  T.Nr4 = v_max * 6 / PI + pow(ds, 3.0) - T.Nr1 - T.Nr2 - T.Nr3;
  if (T.Nr4 <= 0.0) {
    T.d34 = 0;
    T.Nr4 = 0;
  } else {
    T.d34 = exp(log(T.Nr4) / 3);
  }
  T.Nrt = T.Nr1 + T.Nr2 + T.Nr3 + T.Nr4;
  double diameter;
  if (T.Nrt == 0.0) {
    diameter = 0.0;
  } else if (rand < T.Nr1 / T.Nrt) {
    const double X = rand * T.Nrt / T.Nr1;
    diameter = exp(mu1 + sqrt(2.0) * sigma1 * erfinv_approx(X, VA, Vb));
  } else if (rand < (T.Nr2 + T.Nr1) / T.Nrt) {
    const double X = (rand * T.Nrt - T.Nr1) / T.Nr2;
    diameter = mu2 + sqrt(2.0) * sigma2 * erfinv_approx(X, VA, Vb);
  } else if (rand < (T.Nr3 + T.Nr2 + T.Nr1) / T.Nrt) {
    const double X = (rand * T.Nrt - T.Nr1 - T.Nr2) / T.Nr3;
    diameter = mu3 + sqrt(2.0) * sigma3 * erfinv_approx(X, VA, Vb);
  } else {
    diameter = T.d34;
  }
  return pow(diameter, 3.0) * PI / 6;
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

// :62-93
__device__ __forceinline__ void compute_transfer_multiplicities(
    double gamma, int64_t nj, int64_t nk, double mj, double mk, double fragment_mass_i,
    int64_t max_multiplicity, double &take_from_j, double &new_mult_k, int64_t &gamma_j_k,
    bool &overflow) {
  overflow = false;
  gamma_j_k = 0;
  double take_from_j_test = (double)nk;
  take_from_j = 0;
  double new_mult_k_test = ((mj + mk) / fragment_mass_i) * (double)nk;
  new_mult_k = (double)nk;
  const int64_t g = (int64_t)gamma;
  for (int64_t m = 0; m < g; ++m) {
    if (new_mult_k_test > (double)max_multiplicity) { overflow = true; break; }
    if (take_from_j_test > (double)nj) break;
    take_from_j = take_from_j_test;
    new_mult_k = new_mult_k_test;
    gamma_j_k = m + 1;
    take_from_j_test += new_mult_k_test;
    new_mult_k_test = new_mult_k_test * (mj / fragment_mass_i) + new_mult_k_test;
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
