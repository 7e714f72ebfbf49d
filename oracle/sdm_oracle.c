/*
 * sdm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, serial, strict-IEEE (compile with -ffp-contract=off, no -ffast-math) restatement of
 * the algorithm of the reference's SDM collision hot path (jtbuch/PySDM, Numba CPU backend
 * semantics).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Each function cites the reference file:line (relative to the reference root) it follows.
 *
 * Parity pin: checked against golden vectors produced by running the reference itself in its
 * pure-Python mode (tests/golden/gen_golden.py -> tests/golden/ *.npz) and against the
 * known-answer tables of the reference's own unit tests (re-typed in tests/test_oracle_*.py).
 *
 * Conventions: int64 == Storage.INT, double == Storage.FLOAT, uint8 == Storage.BOOL
 * (PySDM/backends/impl_numba/storage.py:16-19).  All pointers are host pointers.
 */
#include <math.h>
/* one implementation of pow / exp / log / erf / sinh / asinh / atanh for the checker and the
 * product (see the header: same bits on the host and on the GPU) */
#include "../pysdm_amd/csrc/sdm_math.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* OpenMP (the `_omp` build, bench.py's cpu_baseline at several threads): the loops the reference's
 * Numba backend runs under `prange` carry `omp parallel for`; what it runs serially stays serial
 * (shuffle within a cell, the middle loop of the adaptive scaling, compaction, counting sort).
 * Counters are integer atomics: results do not depend on the thread count.  Without -fopenmp the
 * pragmas are ignored and this file is the serial checker. */

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * a-1  RNG: NumPy PCG64 (XSL-RR 128/64), numpy/random/src/pcg64/pcg64.h (numpy 2.2.6; third
 * party, not in /root/reference).  Call sites: PySDM/backends/impl_numba/random.py:13-19
 * (`generator.uniform(0, 1, shape)`); stream layout: PySDM/dynamics/impl/
 * random_generator_optimizer.py:19-48.  state[0..1] = LCG state hi,lo; state[2..3] = inc hi,lo
 * (as given by numpy.random.PCG64(seed).state).
 * ---------------------------------------------------------------------------------------- */
static const u128 PCG_MULT = (((u128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;

static inline uint64_t rotr64(uint64_t v, unsigned r) {
  return (v >> r) | (v << ((-r) & 63));
}

API void oracle_pcg64_fill(uint64_t st[4], double *out, int64_t n) {
  u128 state = (((u128)st[0]) << 64) | st[1];
  const u128 inc = (((u128)st[2]) << 64) | st[3];
  for (int64_t i = 0; i < n; ++i) {
    state = state * PCG_MULT + inc;
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const uint64_t r = rotr64(hi ^ lo, (unsigned)(hi >> 58));
    out[i] = (double)(r >> 11) * (1.0 / 9007199254740992.0);
  }
  st[0] = (uint64_t)(state >> 64);
  st[1] = (uint64_t)state;
}

/* jump-ahead by `delta` draws (pcg_advance_lcg_128) */
API void oracle_pcg64_advance(uint64_t st[4], uint64_t delta_hi, uint64_t delta_lo) {
  u128 state = (((u128)st[0]) << 64) | st[1];
  const u128 inc = (((u128)st[2]) << 64) | st[3];
  u128 delta = (((u128)delta_hi) << 64) | delta_lo;
  u128 acc_mult = 1, acc_plus = 0, cur_mult = PCG_MULT, cur_plus = inc;
  while (delta > 0) {
    if (delta & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta >>= 1;
  }
  state = acc_mult * state + acc_plus;
  st[0] = (uint64_t)(state >> 64);
  st[1] = (uint64_t)state;
}

/* ------------------------------------------------------------------------------------------
 * a-2/a-3  index methods, PySDM/backends/impl_numba/methods/index_methods.py
 * ---------------------------------------------------------------------------------------- */
API void oracle_identity_index(int64_t *idx, int64_t n) { /* :14-20 */
  for (int64_t i = 0; i < n; ++i) idx[i] = i;
}

API void oracle_shuffle_global(int64_t *idx, int64_t length, const double *u01) { /* :22-29 */
  for (int64_t i = length - 1; i > 0; --i) {
    const int64_t j = (int64_t)(u01[i] * (double)(i + 1));
    const int64_t t = idx[i];
    idx[i] = idx[j];
    idx[j] = t;
  }
}

API void oracle_shuffle_local(int64_t *idx, const double *u01, const int64_t *cell_start,
                              int64_t n_cell) { /* :32-43 ; prange over the cells */
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t c = 0; c < n_cell; ++c) {
    for (int64_t i = cell_start[c + 1] - 1; i > cell_start[c]; --i) {
      const int64_t j = (int64_t)((double)cell_start[c] +
                                  u01[i] * (double)(cell_start[c + 1] - cell_start[c]));
      const int64_t t = idx[i];
      idx[i] = idx[j];
      idx[j] = t;
    }
  }
}

/* :46-48  idx[:] = argsort(keys, kind="stable")[::-1] */
API void oracle_sort_by_key(int64_t *idx, const double *keys, int64_t n) {
  /* stable ascending insertion ranks, then reversed (n is the number of cells: small) */
  for (int64_t i = 0; i < n; ++i) {
    int64_t rank = 0;
    for (int64_t j = 0; j < n; ++j)
      if (keys[j] < keys[i] || (keys[j] == keys[i] && j < i)) ++rank;
    idx[n - 1 - rank] = i;
  }
}

/* ------------------------------------------------------------------------------------------
 * a-18  PySDM/backends/impl_numba/methods/collisions_methods.py:664-680
 * ---------------------------------------------------------------------------------------- */
API int64_t oracle_remove_zero_n_or_flagged(const int64_t *multiplicity, int64_t *idx,
                                            int64_t length, int64_t idx_len) {
  const int64_t flag = idx_len;
  int64_t new_length = length, i = 0;
  while (i < new_length) {
    if (idx[i] == flag || multiplicity[idx[i]] == 0) {
      new_length -= 1;
      idx[i] = idx[new_length];
      idx[new_length] = flag;
    } else {
      i += 1;
    }
  }
  return new_length;
}

/* a-4  collisions_methods.py:682-697 (serial counting sort; cell_start has n_cell+1 entries) */
API void oracle_counting_sort_by_cell_id(int64_t *new_idx, const int64_t *idx,
                                         const int64_t *cell_id, const int64_t *cell_idx,
                                         int64_t length, int64_t *cell_start,
                                         int64_t cell_start_len) {
  int64_t *cell_end = cell_start;
  for (int64_t i = 0; i < cell_start_len; ++i) cell_end[i] = 0;
  for (int64_t i = 0; i < length; ++i) cell_end[cell_idx[cell_id[idx[i]]]] += 1;
  for (int64_t i = 1; i < cell_start_len; ++i) cell_end[i] += cell_end[i - 1];
  for (int64_t i = length - 1; i >= 0; --i) {
    const int64_t c = cell_idx[cell_id[idx[i]]];
    cell_end[c] -= 1;
    new_idx[cell_end[c]] = idx[i];
  }
}

/* ------------------------------------------------------------------------------------------
 * a-5..a-7  PySDM/backends/impl_numba/methods/pair_methods.py
 * ---------------------------------------------------------------------------------------- */
API void oracle_find_pairs(const int64_t *cell_start, uint8_t *is_first_in_pair,
                           const int64_t *cell_id, const int64_t *cell_idx,
                           const int64_t *idx, int64_t length) { /* :34-55 */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length - 1; ++i) {
    const int same = cell_id[idx[i]] == cell_id[idx[i + 1]];
    const int64_t d = i - cell_start[cell_idx[cell_id[idx[i]]]];
    /* Python % : sign of the divisor, so (d % 2 == 0) iff d is even (also for d < 0) */
    const int even = (d % 2) == 0;
    is_first_in_pair[i] = (uint8_t)(same && even);
  }
  if (length >= 1) is_first_in_pair[length - 1] = 0; /* numpy index -1 when length==0: n/a */
}

API void oracle_sort_within_pair_by_attr_i64(int64_t *idx, int64_t length,
                                             const uint8_t *flag,
                                             const int64_t *attr) { /* :126-140 */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length - 1; ++i)
    if (flag[i] && attr[idx[i]] < attr[idx[i + 1]]) {
      const int64_t t = idx[i];
      idx[i] = idx[i + 1];
      idx[i + 1] = t;
    }
}

API void oracle_sort_within_pair_by_attr_f64(int64_t *idx, int64_t length,
                                             const uint8_t *flag, const double *attr) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length - 1; ++i)
    if (flag[i] && attr[idx[i]] < attr[idx[i + 1]]) {
      const int64_t t = idx[i];
      idx[i] = idx[i + 1];
      idx[i + 1] = t;
    }
}

/* op: 0 sum (:142-160) 1 max (:57-75) 2 min (:77-95) 3 distance (:14-32) 4 multiply (:162-180)
 * `data_out[:] = 0` over the FULL pair array (n_out), then flagged pairs only.            */
static inline double pair_op(int op, double a, double b) {
  switch (op) {
    case 0: return a + b;
    case 1: return a > b ? a : b; /* Python max(a, b): b if b > a else a -- same for non-NaN */
    case 2: return a < b ? a : b;
    case 3: return fabs(a - b);
    default: return a * b;
  }
}

API void oracle_pair_op_f64(int op, double *out, int64_t n_out, const double *in,
                            const uint8_t *flag, const int64_t *idx, int64_t length) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_out; ++i) out[i] = 0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length - 1; ++i)
    if (flag[i]) out[i / 2] = pair_op(op, in[idx[i]], in[idx[i + 1]]);
}

API void oracle_pair_op_i64(int op, double *out, int64_t n_out, const int64_t *in,
                            const uint8_t *flag, const int64_t *idx, int64_t length) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_out; ++i) out[i] = 0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length - 1; ++i)
    if (flag[i]) {
      const int64_t a = in[idx[i]], b = in[idx[i + 1]];
      int64_t r;
      switch (op) {
        case 0: r = a + b; break;
        case 1: r = a > b ? a : b; break;
        case 2: r = a < b ? a : b; break;
        case 3: r = a > b ? a - b : b - a; break;
        default: r = a * b; break;
      }
      out[i / 2] = (double)r;
    }
}

/* :97-124 sort_pair: out has n_sd entries */
API void oracle_sort_pair_f64(double *out, int64_t n_out, const double *in, const uint8_t *flag,
                              const int64_t *idx, int64_t length) {
  for (int64_t i = 0; i < n_out; ++i) out[i] = 0;
  for (int64_t i = 0; i < length - 1; ++i)
    if (flag[i]) {
      const double a = in[idx[i]], b = in[idx[i + 1]];
      if (a < b) { out[i] = b; out[i + 1] = a; } else { out[i] = a; out[i + 1] = b; }
    }
}

/* ------------------------------------------------------------------------------------------
 * a-10  normalize, collisions_methods.py:633-662
 * ---------------------------------------------------------------------------------------- */
API void oracle_normalize(double *prob, int64_t n_prob, const int64_t *cell_id,
                          const int64_t *cell_idx, const int64_t *cell_start,
                          double *norm_factor, int64_t n_cell, double timestep, double dv) {
  for (int64_t i = 0; i < n_cell; ++i) {
    const int64_t sd_num = cell_start[i + 1] - cell_start[i];
    if (sd_num < 2)
      norm_factor[i] = 0;
    else /* left-to-right; int*int stays int in Python before the float division chain */
      norm_factor[i] = timestep / dv * (double)sd_num * (double)(sd_num - 1) / 2 /
                       (double)(sd_num / 2);
  }
  /* NB: cell_id is indexed by the PAIR index d (raw SD #d) -- reference quirk, replicated */
#pragma omp parallel for schedule(static)
  for (int64_t d = 0; d < n_prob; ++d) prob[d] *= norm_factor[cell_idx[cell_id[d]]];
}

/* a-13  pair_indices, collisions_methods.py:16-35 */
static inline int pair_indices(int64_t i, const int64_t *idx, const uint8_t *flag,
                               const double *prob_like, int64_t *j, int64_t *k) {
  if (prob_like[i] == 0) { *j = -1; *k = -1; return 1; }
  const int64_t offset = 1 - (int64_t)flag[2 * i];
  *j = idx[2 * i + offset];
  *k = idx[2 * i + 1 + offset];
  return 0;
}

/* a-11  scale_prob_for_adaptive_sdm_gamma, collisions_methods.py:330-405 */
API void oracle_scale_prob_for_adaptive_sdm_gamma(
    double *prob, const int64_t *idx, int64_t length, const int64_t *multiplicity,
    const int64_t *cell_id, double *dt_left, int64_t n_cell, double dt, double dt_min,
    double dt_max, const uint8_t *flag, int64_t *stats_n_substep, double *stats_dt_min) {
  double *dt_todo = (double *)malloc(sizeof(double) * (size_t)(n_cell > 0 ? n_cell : 1));
  /* Python min(a, b) = b if b < a else a */
  for (int64_t c = 0; c < n_cell; ++c) dt_todo[c] = dt_max < dt_left[c] ? dt_max : dt_left[c];
  for (int64_t i = 0; i < length / 2; ++i) {
    int64_t j, k;
    if (pair_indices(i, idx, flag, prob, &j, &k)) continue;
    const int64_t prop = multiplicity[j] / multiplicity[k];
    double dt_optimal = dt * (double)prop / prob[i];
    const int64_t cid = cell_id[j];
    /* Python max(a, b) = b if b > a else a (a NaN dt_min -- used by the reference's tests --
     * leaves dt_optimal unclamped) */
    dt_optimal = dt_min > dt_optimal ? dt_min : dt_optimal;
    dt_todo[cid] = dt_todo[cid] < dt_optimal ? dt_todo[cid] : dt_optimal;
    /* Python min(a, b) = b if b < a else a: a NaN stats_dt_min (initial fill) stays NaN */
    stats_dt_min[cid] = dt_optimal < stats_dt_min[cid] ? dt_optimal : stats_dt_min[cid];
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length / 2; ++i) {
    int64_t j, k;
    if (pair_indices(i, idx, flag, prob, &j, &k)) continue;
    prob[i] *= dt_todo[cell_id[j]] / dt;
  }
  for (int64_t c = 0; c < n_cell; ++c) {
    dt_left[c] -= dt_todo[c];
    if (dt_todo[c] > 0) stats_n_substep[c] += 1;
  }
  free(dt_todo);
}

/* a-12  compute_gamma, collisions_methods.py:522-585 (out may alias prob) */
API void oracle_compute_gamma(const double *prob, const double *rand, const int64_t *idx,
                              int64_t length, const int64_t *multiplicity,
                              const int64_t *cell_id, int64_t *collision_rate_deficit,
                              int64_t *collision_rate, const uint8_t *flag, double *out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length / 2; ++i) {
    out[i] = ceil(prob[i] - rand[i]);
    int64_t j, k;
    if (pair_indices(i, idx, flag, out, &j, &k)) continue;
    const int64_t prop = multiplicity[j] / multiplicity[k];
    const int64_t gi = (int64_t)out[i];
    const int64_t g = gi < prop ? gi : prop;
    const int64_t cid = cell_id[j];
    const int64_t hit = g * multiplicity[k], missed = (gi - g) * multiplicity[k];
#pragma omp atomic
    collision_rate[cid] += hit;
#pragma omp atomic
    collision_rate_deficit[cid] += missed;
    out[i] = (double)g;
  }
}

/* a-19  adaptive_sdm_end, collisions_methods.py:313-328 */
API int64_t oracle_adaptive_sdm_end(const double *dt_left, int64_t n_cell,
                                    const int64_t *cell_start) {
  int64_t end = 0;
  for (int64_t i = n_cell - 1; i >= 0; --i) {
    if (dt_left[i] == 0) continue;
    end = cell_start[i + 1];
    break;
  }
  return end;
}

/* a-14  coalesce, collisions_methods.py:44-59; attributes is (n_attr, n_sd) row-major */
static inline void coalesce(int64_t i, int64_t j, int64_t k, int64_t cid, int64_t *multiplicity,
                            const double *gamma, double *attributes, int64_t n_attr,
                            int64_t n_sd, int64_t *coalescence_rate) {
  /* atomic_add(int64 array, float) -> in-place add with cast back to int64 */
#ifdef _OPENMP
  {  /* gamma and multiplicities are integer-valued: the same number, added atomically */
    const int64_t add = (int64_t)(gamma[i] * (double)multiplicity[k]);
#pragma omp atomic
    coalescence_rate[cid] += add;
  }
#else
  coalescence_rate[cid] = (int64_t)((double)coalescence_rate[cid] +
                                    gamma[i] * (double)multiplicity[k]);
#endif
  const double new_n = (double)multiplicity[j] - gamma[i] * (double)multiplicity[k];
  if (new_n > 0) {
    multiplicity[j] = (int64_t)new_n;
    for (int64_t a = 0; a < n_attr; ++a)
      attributes[a * n_sd + k] += gamma[i] * attributes[a * n_sd + j];
  } else {
    multiplicity[j] = multiplicity[k] / 2;
    multiplicity[k] = multiplicity[k] - multiplicity[j];
    for (int64_t a = 0; a < n_attr; ++a) {
      attributes[a * n_sd + j] = gamma[i] * attributes[a * n_sd + j] + attributes[a * n_sd + k];
      attributes[a * n_sd + k] = attributes[a * n_sd + j];
    }
  }
}

/* collisions_methods.py:38-41 */
static inline void flag_zero_multiplicity(int64_t j, int64_t k, const int64_t *multiplicity,
                                          int64_t *healthy) {
  if (multiplicity[k] == 0 || multiplicity[j] == 0) healthy[0] = 0;
}

/* collisions_methods.py:418-453 */
API void oracle_collision_coalescence(int64_t *multiplicity, const int64_t *idx, int64_t length,
                                      double *attributes, int64_t n_attr, int64_t n_sd,
                                      const double *gamma, int64_t *healthy,
                                      const int64_t *cell_id, int64_t *coalescence_rate,
                                      const uint8_t *flag) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < length / 2; ++i) {
    int64_t j, k;
    if (pair_indices(i, idx, flag, gamma, &j, &k)) continue;
    coalesce(i, j, k, cell_id[j], multiplicity, gamma, attributes, n_attr, n_sd,
             coalescence_rate);
    flag_zero_multiplicity(j, k, multiplicity, healthy);
  }
}

/* ------------------------------------------------------------------------------------------
 * a-15  breakup, collisions_methods.py:62-311
 * ---------------------------------------------------------------------------------------- */
/* Python round(): half-to-even, result int */
static inline int64_t py_round(double x) { return (int64_t)nearbyint(x); }

/* :62-93 */
static void compute_transfer_multiplicities(double gamma, int64_t j, int64_t k,
                                            const int64_t *multiplicity,
                                            const double *particle_mass, double fragment_mass_i,
                                            int64_t max_multiplicity, double *take_from_j,
                                            double *new_mult_k, int64_t *gamma_j_k,
                                            int *overflow_flag) {
  *overflow_flag = 0;
  *gamma_j_k = 0;
  double take_from_j_test = (double)multiplicity[k];
  *take_from_j = 0;
  double new_mult_k_test =
      ((particle_mass[j] + particle_mass[k]) / fragment_mass_i) * (double)multiplicity[k];
  *new_mult_k = (double)multiplicity[k];
  const int64_t g = (int64_t)gamma;
  for (int64_t m = 0; m < g; ++m) {
    if (new_mult_k_test > (double)max_multiplicity) { *overflow_flag = 1; break; }
    if (take_from_j_test > (double)multiplicity[j]) break;
    *take_from_j = take_from_j_test;
    *new_mult_k = new_mult_k_test;
    *gamma_j_k = m + 1;
    take_from_j_test += new_mult_k_test;
    new_mult_k_test = new_mult_k_test * (particle_mass[j] / fragment_mass_i) + new_mult_k_test;
  }
}

/* :96-114 */
static void get_new_multiplicities_and_update_attributes(int64_t j, int64_t k, double *attributes,
                                                         int64_t n_attr, int64_t n_sd,
                                                         const int64_t *multiplicity,
                                                         double take_from_j, double new_mult_k,
                                                         double *nj, double *nk) {
  for (int64_t a = 0; a < n_attr; ++a) {
    attributes[a * n_sd + k] *= (double)multiplicity[k];
    attributes[a * n_sd + k] += take_from_j * attributes[a * n_sd + j];
    attributes[a * n_sd + k] /= new_mult_k;
  }
  if ((double)multiplicity[j] > take_from_j) {
    *nj = (double)multiplicity[j] - take_from_j;
    *nk = new_mult_k;
  } else {
    *nj = new_mult_k / 2;
    *nk = *nj;
    for (int64_t a = 0; a < n_attr; ++a) attributes[a * n_sd + j] = attributes[a * n_sd + k];
  }
}

/* :117-132 */
static void round_multiplicities_to_ints_and_update_attributes(int64_t j, int64_t k, double nj,
                                                               double nk, double *attributes,
                                                               int64_t n_attr, int64_t n_sd,
                                                               int64_t *multiplicity) {
  int64_t rj = py_round(nj), rk = py_round(nk);
  multiplicity[j] = rj > 1 ? rj : 1;
  multiplicity[k] = rk > 1 ? rk : 1;
  const double factor_j = nj / (double)multiplicity[j];
  const double factor_k = nk / (double)multiplicity[k];
  for (int64_t a = 0; a < n_attr; ++a) {
    attributes[a * n_sd + k] *= factor_k;
    attributes[a * n_sd + j] *= factor_j;
  }
}

static inline void atomic_add_i64_f64(int64_t *arr, int64_t i, double v) {
#ifdef _OPENMP
  const int64_t add = (int64_t)v; /* integer-valued by construction (counts of droplets) */
#pragma omp atomic
  arr[i] += add;
#else
  arr[i] = (int64_t)((double)arr[i] + v);
#endif
}

/* :135-175 ; returns overflow flag */
static int break_up(int64_t i, int64_t j, int64_t k, int64_t cid, int64_t *multiplicity,
                    const double *gamma, double *attributes, int64_t n_attr, int64_t n_sd,
                    const double *fragment_mass, int64_t max_multiplicity, int64_t *breakup_rate,
                    int64_t *breakup_rate_deficit, const double *particle_mass) {
  double take_from_j, new_mult_k, nj, nk;
  int64_t gamma_j_k;
  int overflow;
  compute_transfer_multiplicities(gamma[i], j, k, multiplicity, particle_mass, fragment_mass[i],
                                  max_multiplicity, &take_from_j, &new_mult_k, &gamma_j_k,
                                  &overflow);
  const double gamma_deficit = gamma[i] - (double)gamma_j_k;
  get_new_multiplicities_and_update_attributes(j, k, attributes, n_attr, n_sd, multiplicity,
                                               take_from_j, new_mult_k, &nj, &nk);
  {
    const int64_t broke = gamma_j_k * multiplicity[k]; /* int * int */
#pragma omp atomic
    breakup_rate[cid] += broke;
  }
  atomic_add_i64_f64(breakup_rate_deficit, cid, gamma_deficit * (double)multiplicity[k]);
  round_multiplicities_to_ints_and_update_attributes(j, k, nj, nk, attributes, n_attr, n_sd,
                                                     multiplicity);
  return overflow;
}

/* :178-243 */
static int break_up_while(int64_t i, int64_t j, int64_t k, int64_t cid, int64_t *multiplicity,
                          const double *gamma, double *attributes, int64_t n_attr, int64_t n_sd,
                          const double *fragment_mass, int64_t max_multiplicity,
                          int64_t *breakup_rate, int64_t *breakup_rate_deficit,
                          const double *particle_mass) {
  double gamma_deficit = gamma[i];
  int overflow = 0;
  while (gamma_deficit > 0) {
    double take_from_j, new_mult_k, nj, nk, gamma_j_k;
    if (multiplicity[k] == multiplicity[j]) {
      take_from_j = (double)multiplicity[j];
      new_mult_k = (particle_mass[j] + particle_mass[k]) / fragment_mass[i] *
                   (double)multiplicity[k];
      if (new_mult_k > (double)max_multiplicity) {
        atomic_add_i64_f64(breakup_rate_deficit, cid, gamma_deficit * (double)multiplicity[k]);
        overflow = 1;
        break;
      }
      gamma_j_k = gamma_deficit;
    } else {
      if (multiplicity[k] > multiplicity[j]) { const int64_t t = j; j = k; k = t; }
      int64_t g_int;
      compute_transfer_multiplicities(gamma_deficit, j, k, multiplicity, particle_mass,
                                      fragment_mass[i], max_multiplicity, &take_from_j,
                                      &new_mult_k, &g_int, &overflow);
      gamma_j_k = (double)g_int;
      /* where not one breakup fits, the reference's loop never ends (gamma_deficit -= 0): the
       * product leaves it (a hung wavefront can take the whole GPU down) and so does the checker;
       * the remainder goes to the deficit below */
      if (g_int == 0) break;
    }
    get_new_multiplicities_and_update_attributes(j, k, attributes, n_attr, n_sd, multiplicity,
                                                 take_from_j, new_mult_k, &nj, &nk);
    atomic_add_i64_f64(breakup_rate, cid, gamma_j_k * (double)multiplicity[k]);
    gamma_deficit -= gamma_j_k;
    round_multiplicities_to_ints_and_update_attributes(j, k, nj, nk, attributes, n_attr, n_sd,
                                                       multiplicity);
  }
  atomic_add_i64_f64(breakup_rate_deficit, cid, gamma_deficit * (double)multiplicity[k]);
  return overflow;
}

/* :247-311 ; returns the number of overflow events (the reference prints a warning each) */
API int64_t oracle_collision_coalescence_breakup(
    int64_t *multiplicity, const int64_t *idx, int64_t length, double *attributes,
    int64_t n_attr, int64_t n_sd, const double *gamma, const double *rand, const double *Ec,
    const double *Eb, const double *fragment_mass, int64_t *healthy, const int64_t *cell_id,
    int64_t *coalescence_rate, int64_t *breakup_rate, int64_t *breakup_rate_deficit,
    const uint8_t *flag, int64_t max_multiplicity, const double *particle_mass,
    int handle_all_breakups) {
  int64_t n_overflow = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_overflow)
  for (int64_t i = 0; i < length / 2; ++i) {
    int64_t j, k;
    if (pair_indices(i, idx, flag, gamma, &j, &k)) continue;
    const int bouncing = rand[i] - (Ec[i] + (1 - Ec[i]) * (Eb[i])) > 0;
    if (bouncing) continue;
    if (rand[i] - Ec[i] < 0) {
      coalesce(i, j, k, cell_id[j], multiplicity, gamma, attributes, n_attr, n_sd,
               coalescence_rate);
    } else if (handle_all_breakups) {
      n_overflow += break_up_while(i, j, k, cell_id[j], multiplicity, gamma, attributes, n_attr,
                                   n_sd, fragment_mass, max_multiplicity, breakup_rate,
                                   breakup_rate_deficit, particle_mass);
    } else {
      n_overflow += break_up(i, j, k, cell_id[j], multiplicity, gamma, attributes, n_attr, n_sd,
                             fragment_mass, max_multiplicity, breakup_rate,
                             breakup_rate_deficit, particle_mass);
    }
    flag_zero_multiplicity(j, k, multiplicity, healthy);
  }
  return n_overflow;
}

/* collisions_methods.py:407-416  cell_id = strides . cell_origin ; cell_origin is (n_dim, n_sd) */
API void oracle_cell_id(int64_t *cell_id, const int64_t *cell_origin, const int64_t *strides,
                        int64_t n_dim, int64_t n_sd) {
  for (int64_t i = 0; i < n_sd; ++i) {
    int64_t s = 0;
    for (int64_t d = 0; d < n_dim; ++d) s += strides[d] * cell_origin[d * n_sd + i];
    cell_id[i] = s;
  }
}

/* ------------------------------------------------------------------------------------------
 * a-8/a-16  Berry-type collection efficiency, collisions_methods.py:743-782
 * ---------------------------------------------------------------------------------------- */
API void oracle_linear_collection_efficiency(const double *params, double *output, int64_t n_out,
                                             const double *radii, const uint8_t *flag,
                                             const int64_t *idx, int64_t length, double unit) {
  const double A = params[0], B = params[1], D1 = params[2], D2 = params[3], E1 = params[4],
               E2 = params[5], F1 = params[6], F2 = params[7], G1 = params[8], G2 = params[9],
               G3 = params[10], Mf = params[11], Mg = params[12];
  for (int64_t i = 0; i < n_out; ++i) output[i] = 0;
  for (int64_t i = 0; i < length - 1; ++i) {
    if (!flag[i]) continue;
    double r, r_s;
    if (radii[idx[i]] > radii[idx[i + 1]]) {
      r = radii[idx[i]] / unit;
      r_s = radii[idx[i + 1]] / unit;
    } else {
      r = radii[idx[i + 1]] / unit;
      r_s = radii[idx[i]] / unit;
    }
    const double p = r_s / r;
    if (p != 0 && p != 1) {
      const double G = sdm_pow(G1 / r, Mg) + G2 + G3 * r;
      const double Gp = sdm_pow(1 - p, G);
      if (Gp != 0) {
        const double D = D1 / sdm_pow(r, D2);
        const double E = E1 / sdm_pow(r, E2);
        const double F = sdm_pow(F1 / r, Mf) + F2;
        double v = A + B * p + D / sdm_pow(p, F) + E / Gp;
        output[i / 2] = v > 0 ? v : 0; /* max(0, v) */
      }
    }
  }
}

/* a-9  PySDM/backends/impl_numba/methods/terminal_velocity_methods.py:14-30 */
/* table_len: beyond the table the last row is used (the reference raises before it gets here,
 * dynamics/terminal_velocity/gunn_and_kinzer.py:127-134; the product's kernels clamp likewise so
 * that no read leaves the table - identical on both sides, tested on runs that grow past it) */
API void oracle_interpolation(double *output, const double *radius, int64_t n, double factor,
                              const double *b, const double *c, int64_t table_len) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    if (radius[i] < 0) {
      output[i] = 0;
    } else {
      const double x = factor * radius[i];
      int64_t r_id = (int64_t)x;
      if (table_len > 0 && r_id > table_len - 1) r_id = table_len - 1;
      const double r_rest = fmod(x, 1.0) / factor; /* x >= 0: Python % == fmod */
      output[i] = b[r_id] + r_rest * c[r_id];
    }
  }
}

/* physics_methods.py:107-131 + physics/particle_shape_and_density/liquid_spheres.py:18-23 */
API void oracle_volume_of_water_mass(double *volume, const double *mass, int64_t n,
                                     double rho_w) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) volume[i] = mass[i] / rho_w;
}
API void oracle_mass_of_water_volume(double *mass, const double *volume, int64_t n,
                                     double rho_w) {
  for (int64_t i = 0; i < n; ++i) mass[i] = rho_w * volume[i];
}

/* ------------------------------------------------------------------------------------------
 * a-17  fragmentation, PySDM/backends/impl_numba/methods/fragmentation_methods.py
 * ---------------------------------------------------------------------------------------- */
/* :76-95 ; nfmax < 0 encodes None */
API void oracle_fragmentation_limiters(double *n_fragment, double *frag_volume, int64_t n,
                                       double vmin, double nfmax, const double *x_plus_y) {
  for (int64_t i = 0; i < n; ++i) {
    if (x_plus_y[i] == 0.0) {
      frag_volume[i] = 0.0;
      n_fragment[i] = 1.0;
    } else {
      if (isnan(frag_volume[i]) || frag_volume[i] == 0.0) frag_volume[i] = x_plus_y[i];
      frag_volume[i] = frag_volume[i] < x_plus_y[i] ? frag_volume[i] : x_plus_y[i];
      if (nfmax >= 0 && x_plus_y[i] / frag_volume[i] > nfmax)
        frag_volume[i] = x_plus_y[i] / nfmax;
      else if (frag_volume[i] < vmin)
        frag_volume[i] = x_plus_y[i];
      n_fragment[i] = x_plus_y[i] / frag_volume[i];
    }
  }
}

/* PySDM/physics/trivia.py:95-108 (Vedder 1987) */
static inline double erfinv_approx(double c, double VA, double Vb) {
  return 2 * sqrt(VA) * sdm_sinh(sdm_asinh(sdm_atanh(c) / 2 / Vb / sdm_pow(VA, 1.5)) / 3);
}

/* :477-485 gauss ; consts = {VEDDER_1987_A, VEDDER_1987_b} */
API void oracle_gauss_fragmentation(double mu, double sigma, double *frag_volume,
                                    const double *rand, int64_t n, const double *consts) {
  for (int64_t i = 0; i < n; ++i)
    frag_volume[i] = mu + sigma * erfinv_approx(rand[i], consts[0], consts[1]);
}

/* :487-499 + PySDM/physics/fragmentation_function/feingold1988.py:13-15 */
API void oracle_feingold1988_fragmentation(double scale, double *frag_volume,
                                           const double *x_plus_y, const double *rand, int64_t n,
                                           double fragtol) {
  for (int64_t i = 0; i < n; ++i) {
    const double a = 1 - rand[i] * scale / x_plus_y[i];
    frag_volume[i] = -scale * sdm_log(a > fragtol ? a : fragtol);
  }
}


/* Python's max(a, b) = b if b > a else a ; min(a, b) = b if b < a else a */
#define PYMAX(a, b) ((b) > (a) ? (b) : (a))
#define PYMIN(a, b) ((b) < (a) ? (b) : (a))
#define LL_PI 3.141592653589793

/* Low & List 1982 fragment-size parameters, PySDM/physics/fragmentation_function/lowlist82.py
 * (each returns H, mu, sigma in cm units; the fixed-point loops run 10 rounds as there) */
struct LL82P { double H, mu, sigma; };

static struct LL82P ll82_gauss_fixed_point(double H, double mu, double upper) {
  /* :22-30, :108-116, :150-158: sigma <- sqrt(2/pi)/H / (1 + sdm_erf((upper - mu)/(sqrt2 sigma))) */
  double sigma = 1 / H;
  for (int r = 0; r < 10; ++r)
    sigma = 1 / H * sqrt(2 / LL_PI) / (1 + sdm_erf((upper - mu) / (sqrt(2.0) * sigma)));
  struct LL82P p = {H, mu, sigma};
  return p;
}

static struct LL82P ll82_f1(double CM, double dl, double dcoal) { /* :15-30 */
  const double dlCM = dl / CM;
  return ll82_gauss_fixed_point(50.8 * sdm_pow(dlCM, -0.718), dlCM, dcoal / CM);
}

static struct LL82P ll82_f2(double CM, double ds) { /* :33-38 */
  const double dsCM = ds / CM;
  const double H = 4.18 * sdm_pow(dsCM, -1.17);
  struct LL82P p = {H, dsCM, 1 / (sqrt(2 * LL_PI) * H)};
  return p;
}

static struct LL82P ll82_f3(double CM, double ds, double dl) { /* :41-98 */
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

static struct LL82P ll82_s1(double CM, double dl, double ds, double dcoal) { /* :100-116 */
  return ll82_gauss_fixed_point(100 * sdm_exp(-3.25 * (ds / CM)), dl / CM, dcoal / CM);
}

static struct LL82P ll82_s2(double CM, double dl, double ds, double St) { /* :118-143 */
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

static struct LL82P ll82_d1(double CM, double W1, double dl, double dcoal, double CKE) {
  /* :145-160 */
  const double mu = (dl / CM) * (1 - sdm_exp(-3.70 * (3.10 - W1)));
  return ll82_gauss_fixed_point(1.58e-5 * sdm_pow(CKE, -1.22), mu, dcoal / CM);
}

static struct LL82P ll82_d2(double CM, double ds, double dl, double CKE) { /* :162-193 */
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
static double ll82_fragment_volume(double CKE, double W, double W2, double St, double ds, double dl,
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

/* fragmentation_methods.py:379-474 */
API void oracle_ll82_fragmentation(const double *CKE, const double *W, const double *W2,
                                   const double *St, const double *ds, const double *dl,
                                   const double *dcoal, double *frag_volume, double *rand,
                                   double *Rf, double *Rs, double *Rd, int64_t n, double tol,
                                   const double *consts) {
  for (int64_t i = 0; i < n; ++i)
    frag_volume[i] = ll82_fragment_volume(CKE[i], W[i], W2[i], St[i], ds[i], dl[i], dcoal[i],
                                          &rand[i], &Rf[i], &Rs[i], &Rd[i], tol, consts);
}

/* test hook: the seven parameter triples of lowlist82.py, selected by name index
 * 0 f1(dl,dcoal) 1 f2(ds) 2 f3(ds,dl) 3 s1(dl,ds,dcoal) 4 s2(dl,ds,St) 5 d1(W1,dl,dcoal,CKE)
 * 6 d2(ds,dl,CKE) -- pinned by tests/unit_tests/physics/test_fragmentation_functions.py:76-173 */
API void oracle_ll82_params(int which, double a, double b, double c, double d, double CM,
                            double *out) {
  struct LL82P p = {0, 0, 0};
  switch (which) {
    case 0: p = ll82_f1(CM, a, b); break;
    case 1: p = ll82_f2(CM, a); break;
    case 2: p = ll82_f3(CM, a, b); break;
    case 3: p = ll82_s1(CM, a, b, c); break;
    case 4: p = ll82_s2(CM, a, b, c); break;
    case 5: p = ll82_d1(CM, a, b, c, d); break;
    case 6: p = ll82_d2(CM, a, b, c); break;
    default: break;
  }
  out[0] = p.H; out[1] = p.mu; out[2] = p.sigma;
}

/* fragmentation_methods.py:305-319 */
API void oracle_ll82_coalescence_check(double *Ec, const double *dl, int64_t n) {
  for (int64_t i = 0; i < n; ++i)
    if (dl[i] < 0.4e-3) Ec[i] = 1.0;
}

/* :98-112 slams */
API void oracle_slams_fragmentation(double *n_fragment, double *frag_volume,
                                    const double *x_plus_y, double *probs, const double *rand,
                                    int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    probs[i] = 0.0;
    n_fragment[i] = 1;
    for (int k = 0; k < 22; ++k) {
      probs[i] += 0.91 * sdm_pow((double)(k + 2), -1.56);
      if (rand[i] < probs[i]) { n_fragment[i] = k + 2; break; }
    }
    frag_volume[i] = x_plus_y[i] / n_fragment[i];
  }
}

/* :136-144 */
API void oracle_exp_fragmentation(double scale, double *frag_volume, const double *rand,
                                  int64_t n, double tol) {
  for (int64_t i = 0; i < n; ++i) {
    const double a = 1 - rand[i];
    frag_volume[i] = -scale * sdm_log(a > tol ? a : tol);
  }
}

/* :321-377 with helpers :12-48 and PySDM/physics/fragmentation_function/straub2010nf.py:13-45.
 * consts = {CM, STRAUB_E_D1, STRAUB_MU2, VEDDER_1987_A, VEDDER_1987_b, PI}                   */
API void oracle_straub_fragmentation(const double *CW, const double *gam, const double *ds,
                                     const double *v_max, double *frag_volume,
                                     const double *rand, double *Nr1, double *Nr2, double *Nr3,
                                     double *Nr4, double *Nrt, double *d34, int64_t n,
                                     const double *consts) {
  const double CM = consts[0], E_D1 = consts[1], MU2 = consts[2], VA = consts[3],
               Vb = consts[4], PI = consts[5];
  for (int64_t i = 0; i < n; ++i) {
    /* straub_Nr :12-32 */
    if (gam[i] * CW[i] >= 7.0) Nr1[i] = 0.088 * (gam[i] * CW[i] - 7.0);
    if (CW[i] >= 21.0) {
      Nr2[i] = 0.22 * (CW[i] - 21.0);
      if (CW[i] <= 46.0) Nr3[i] = 0.04 * (46.0 - CW[i]);
    } else {
      Nr3[i] = 1.0;
    }
    Nr4[i] = 1.0;
    Nrt[i] = Nr1[i] + Nr2[i] + Nr3[i] + Nr4[i];
    /* params straub2010nf.py:13-45 */
    const double sigma1 = sqrt(sdm_log(CW[i] / 64 / 100 * CM * CM / 12 / sdm_pow(E_D1, 2) + 1));
    const double mu1 = sdm_log(E_D1) - sdm_pow(sigma1, 2) / 2;
    const double s2a = 7 * (CW[i] - 21) * CM / 1000;
    const double sigma2 = (s2a > 0.0 ? s2a : 0.0) / sqrt(12.0);
    const double mu2 = MU2;
    const double sigma3 = (1 + 0.76 * sqrt(CW[i])) * CM / 100 / sqrt(12.0);
    const double mu3 = 0.9 * ds[i];
    /* straub_mass_remainder :35-48 */
    Nr1[i] = Nr1[i] * sdm_exp(3 * mu1 + 9 * sdm_pow(sigma1, 2) / 2);
    Nr2[i] = Nr2[i] * (sdm_pow(mu2, 3.0) + 3 * mu2 * sdm_pow(sigma2, 2.0)); /* Python ** -> libm pow */
    Nr3[i] = Nr3[i] * (sdm_pow(mu3, 3.0) + 3 * mu3 * sdm_pow(sigma3, 2.0));
    Nr4[i] = v_max[i] * 6 / PI + sdm_pow(ds[i], 3.0) - Nr1[i] - Nr2[i] - Nr3[i];
    if (Nr4[i] <= 0.0) {
      d34[i] = 0;
      Nr4[i] = 0;
    } else {
      d34[i] = sdm_exp(sdm_log(Nr4[i]) / 3);
    }
    Nrt[i] = Nr1[i] + Nr2[i] + Nr3[i] + Nr4[i];
    double diameter;
    if (Nrt[i] == 0.0) {
      diameter = 0.0;
    } else if (rand[i] < Nr1[i] / Nrt[i]) {
      const double X = rand[i] * Nrt[i] / Nr1[i];
      diameter = sdm_exp(mu1 + sqrt(2.0) * sigma1 * erfinv_approx(X, VA, Vb));
    } else if (rand[i] < (Nr2[i] + Nr1[i]) / Nrt[i]) {
      const double X = (rand[i] * Nrt[i] - Nr1[i]) / Nr2[i];
      diameter = mu2 + sqrt(2.0) * sigma2 * erfinv_approx(X, VA, Vb);
    } else if (rand[i] < (Nr3[i] + Nr2[i] + Nr1[i]) / Nrt[i]) {
      const double X = (rand[i] * Nrt[i] - Nr1[i] - Nr2[i]) / Nr3[i];
      diameter = mu3 + sqrt(2.0) * sigma3 * erfinv_approx(X, VA, Vb);
    } else {
      diameter = d34[i];
    }
    frag_volume[i] = sdm_pow(diameter, 3.0) * PI / 6;
  }
}

/* terminal_velocity_methods.py:32-45 with physics/terminal_velocity/rogers_yau.py:14-25;
 * consts = {SMALL_K, MEDIUM_K, LARGE_K, SMALL_R_LIMIT, MEDIUM_R_LIMIT} */
API void oracle_terminal_velocity(double *values, const double *radius, int64_t n,
                                  const double *consts) {
  for (int64_t i = 0; i < n; ++i) {
    const double r = radius[i];
    values[i] = r < consts[3] ? consts[0] * (r * r)
                              : (r < consts[4] ? consts[1] * r : consts[2] * sdm_pow(r, 0.5));
  }
}

/* terminal_velocity_methods.py:47-66 */
API void oracle_power_series(double *values, const double *radius, int64_t n, int num_terms,
                             const double *prefactors, const double *powers) {
  for (int64_t i = 0; i < n; ++i) {
    values[i] = 0.0;
    for (int j = 0; j < num_terms; ++j)
      values[i] = values[i] + prefactors[j] * sdm_pow(radius[i], powers[j] * 3);
  }
}

/* ------------------------------------------------------------------------------------------
 * f-3  displacement, PySDM/backends/impl_numba/methods/displacement_methods.py
 * ---------------------------------------------------------------------------------------- */
/* :14-129 + physics/particle_advection/{implicit,explicit}_in_space.py: Arakawa-C Courant
 * numbers left and right of the droplet in direction `dim`, interpolated to its position.
 * courant has the grid's shape with one more point along `dim`; scheme 0 implicit, 1 explicit */
API void oracle_calculate_displacement(int dim, int n_dims, int scheme, double *displacement,
                                       const double *courant, const int64_t *courant_shape,
                                       const int64_t *cell_origin, const double *position_in_cell,
                                       int64_t n_sd, double n_substeps) {
  for (int64_t droplet = 0; droplet < n_sd; ++droplet) {
    int64_t l = 0, r = 0;
    for (int d = 0; d < n_dims; ++d) {
      const int64_t o = cell_origin[d * n_sd + droplet];
      l = l * courant_shape[d] + o;
      r = r * courant_shape[d] + o + (d == dim);
    }
    const double x = position_in_cell[dim * n_sd + droplet];
    const double c_l = courant[l] / n_substeps, c_r = courant[r] / n_substeps;
    double v = c_l * (1 - x) + c_r * x;
    if (scheme == 0) v = v / (1 - c_r + c_l);
    displacement[dim * n_sd + droplet] = v;
  }
}

/* :131-166, :192-218: SDs that fall through the counting level are flagged out (idx = n_sd),
 * returns the mass of water they carried (summed in idx order) */
API double oracle_flag_precipitated(const int64_t *cell_origin, const double *position_in_cell,
                                    const double *water_mass, const int64_t *multiplicity,
                                    int64_t *idx, int64_t length, int64_t n_sd, int n_dims,
                                    int64_t *healthy, double level, const double *displacement) {
  double rainfall_mass = 0.0;
  const int64_t last = (int64_t)(n_dims - 1) * n_sd;
  for (int64_t i = 0; i < length; ++i) {
    const int64_t k = idx[i];
    const double z = (double)cell_origin[last + k] + position_in_cell[last + k];
    if (displacement[last + k] < 0 && z < level) {
      rainfall_mass += fabs(water_mass[k]) * (double)multiplicity[k];
      idx[i] = n_sd;
      healthy[0] = 0;
    }
  }
  return rainfall_mass;
}

/* :168-190, :220-238 */
API void oracle_flag_out_of_column(const int64_t *cell_origin, const double *position_in_cell,
                                   int64_t *idx, int64_t length, int64_t n_sd, int n_dims,
                                   int64_t *healthy, double top) {
  const int64_t last = (int64_t)(n_dims - 1) * n_sd;
  for (int64_t i = 0; i < length; ++i) {
    const int64_t k = idx[i];
    const double z = (double)cell_origin[last + k] + position_in_cell[last + k];
    if (z < 0 || z > top) {
      idx[i] = n_sd;
      healthy[0] = 0;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * f-1  moments, PySDM/backends/impl_numba/methods/moments_methods.py:14-99 (serial order)
 * ---------------------------------------------------------------------------------------- */
API void oracle_moments(double *moment_0, double *moments, const int64_t *multiplicity,
                        const double *attr_data, const int64_t *cell_id, const int64_t *idx,
                        int64_t length, const double *ranks, int64_t n_ranks, int64_t n_cell,
                        double min_x, double max_x, const double *x_attr,
                        const double *weighting_attribute, double weighting_rank,
                        int skip_division_by_m0) {
  for (int64_t c = 0; c < n_cell; ++c) moment_0[c] = 0;
  for (int64_t k = 0; k < n_ranks * n_cell; ++k) moments[k] = 0;
  for (int64_t idx_i = 0; idx_i < length; ++idx_i) {
    const int64_t i = idx[idx_i];
    if (min_x <= x_attr[i] && x_attr[i] < max_x) {
      const double w = (double)multiplicity[i] *
                       (weighting_rank == 0 ? 1.0 : sdm_pow(weighting_attribute[i], weighting_rank));
      moment_0[cell_id[i]] += w;
      for (int64_t k = 0; k < n_ranks; ++k)
        moments[k * n_cell + cell_id[i]] += w * sdm_pow(attr_data[i], ranks[k]);
    }
  }
  if (!skip_division_by_m0)
    for (int64_t c = 0; c < n_cell; ++c)
      for (int64_t k = 0; k < n_ranks; ++k)
        moments[k * n_cell + c] = moment_0[c] != 0 ? moments[k * n_cell + c] / moment_0[c] : 0;
}

/* f-1  spectrum_moments, moments_methods.py:100-147: the first bin k with
 * x_bins[k] <= x < x_bins[k+1] takes the SD; moment_0 and moments are (n_bins, n_cell) */
API void oracle_spectrum_moments(double *moment_0, double *moments, const int64_t *multiplicity,
                                 const double *attr_data, const int64_t *cell_id,
                                 const int64_t *idx, int64_t length, double rank,
                                 const double *x_bins, int64_t n_bins, int64_t n_cell,
                                 const double *x_attr, const double *weighting_attribute,
                                 double weighting_rank) {
  for (int64_t k = 0; k < n_bins * n_cell; ++k) moment_0[k] = moments[k] = 0;
  for (int64_t idx_i = 0; idx_i < length; ++idx_i) {
    const int64_t i = idx[idx_i];
    for (int64_t k = 0; k < n_bins; ++k)
      if (x_bins[k] <= x_attr[i] && x_attr[i] < x_bins[k + 1]) {
        const double w = (double)multiplicity[i] * sdm_pow(weighting_attribute[i], weighting_rank);
        moment_0[k * n_cell + cell_id[i]] += w;
        moments[k * n_cell + cell_id[i]] += w * sdm_pow(attr_data[i], rank);
        break;
      }
  }
  for (int64_t k = 0; k < n_bins * n_cell; ++k)
    moments[k] = moment_0[k] != 0 ? moments[k] / moment_0[k] : 0;
}
