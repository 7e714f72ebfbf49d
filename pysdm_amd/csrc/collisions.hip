// collisions.hip -- fine-grained backend-method kernels of the collision path: one symbol per
// reference backend method (pair_methods.py, collisions_methods.py, fragmentation_methods.py,
// terminal_velocity_methods.py, physics_methods.py, moments_methods.py).
#include "common.h"
#include "physics.h"

#define GRID1D(n) dim3(grid_for(n)), dim3(SDM_BLOCK), 0, ctx->stream
#define TID() ((int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x)

// ---- find_pairs (pair_methods.py:34-55) --------------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_find_pairs(const int64_t *__restrict__ cell_start, uint8_t *__restrict__ flag,
             const int64_t *__restrict__ cell_id, const int64_t *__restrict__ cell_idx,
             const int64_t *__restrict__ idx, int64_t length) {
  const int64_t i = TID();
  if (i >= length) return;
  if (i == length - 1) { flag[i] = 0; return; }
  const int64_t ca = cell_id[idx[i]], cb = cell_id[idx[i + 1]];
  const int64_t d = i - cell_start[cell_idx[ca]];
  flag[i] = (uint8_t)((ca == cb) && ((d & 1) == 0));
}

extern "C" int sdm_find_pairs(sdm_ctx *ctx, const int64_t *cell_start, uint8_t *flag,
                              const int64_t *cell_id, const int64_t *cell_idx,
                              const int64_t *idx, int64_t length) {
  ARG_TRY(ctx && length >= 0);
  if (length == 0) return SDM_OK;
  ARG_TRY(cell_start && flag && cell_id && cell_idx && idx);
  hipLaunchKernelGGL(k_find_pairs, GRID1D(length), cell_start, flag, cell_id, cell_idx, idx,
                     length);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- sort_within_pair_by_attr (pair_methods.py:126-140) ---------------------------------
template <typename T>
__global__ void __launch_bounds__(SDM_BLOCK)
k_sort_within_pair(int64_t *__restrict__ idx, int64_t length, const uint8_t *__restrict__ flag,
                   const T *__restrict__ attr) {
  const int64_t i = TID();
  if (i >= length - 1 || !flag[i]) return;
  const int64_t a = idx[i], b = idx[i + 1];
  if (attr[a] < attr[b]) { idx[i] = b; idx[i + 1] = a; }
}

extern "C" int sdm_sort_within_pair_by_attr(sdm_ctx *ctx, int64_t *idx, int64_t length,
                                            const uint8_t *flag, const void *attr,
                                            int attr_is_int) {
  ARG_TRY(ctx && length >= 0);
  if (length < 2) return SDM_OK;
  ARG_TRY(idx && flag && attr);
  if (attr_is_int)
    hipLaunchKernelGGL(k_sort_within_pair<int64_t>, GRID1D(length), idx, length, flag,
                       (const int64_t *)attr);
  else
    hipLaunchKernelGGL(k_sort_within_pair<double>, GRID1D(length), idx, length, flag,
                       (const double *)attr);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- pair reductions (pair_methods.py:14-32,57-95,142-180) -------------------------------
// thread per output slot d: zero-fill is folded in (slot d gets the pair starting at 2d or 2d+1)
template <typename T>
__global__ void __launch_bounds__(SDM_BLOCK)
k_pair_op(int op, double *__restrict__ out, int64_t n_out, const T *__restrict__ in,
          const uint8_t *__restrict__ flag, const int64_t *__restrict__ idx, int64_t length) {
  const int64_t d = TID();
  if (d >= n_out) return;
  double r = 0.0;
  int64_t i = -1;
  if (2 * d < length - 1 && flag[2 * d]) i = 2 * d;
  else if (2 * d + 1 < length - 1 && flag[2 * d + 1]) i = 2 * d + 1;
  if (i >= 0) {
    const T a = in[idx[i]], b = in[idx[i + 1]];
    T v;
    switch (op) {
      case SDM_PAIR_SUM: v = a + b; break;
      case SDM_PAIR_MAX: v = a > b ? a : b; break;
      case SDM_PAIR_MIN: v = a < b ? a : b; break;
      case SDM_PAIR_DISTANCE: v = a > b ? a - b : b - a; break;
      default: v = a * b; break;
    }
    r = (double)v;
  }
  out[d] = r;
}

extern "C" int sdm_pair_op(sdm_ctx *ctx, int op, double *out, int64_t n_out, const void *in,
                           int in_is_int, const uint8_t *flag, const int64_t *idx,
                           int64_t length) {
  ARG_TRY(ctx && n_out >= 0 && length >= 0 && op >= 0 && op <= SDM_PAIR_MULTIPLY);
  if (n_out == 0) return SDM_OK;
  ARG_TRY(out && in && flag && idx);
  if (in_is_int)
    hipLaunchKernelGGL(k_pair_op<int64_t>, GRID1D(n_out), op, out, n_out, (const int64_t *)in,
                       flag, idx, length);
  else
    hipLaunchKernelGGL(k_pair_op<double>, GRID1D(n_out), op, out, n_out, (const double *)in,
                       flag, idx, length);
  LAUNCH_CHECK();
  return SDM_OK;
}

// sort_pair (pair_methods.py:97-124): out has one slot per super-droplet position
__global__ void __launch_bounds__(SDM_BLOCK)
k_sort_pair(double *__restrict__ out, int64_t n_out, const double *__restrict__ in,
            const uint8_t *__restrict__ flag, const int64_t *__restrict__ idx, int64_t length) {
  const int64_t p = TID();
  if (p >= n_out) return;
  double r = 0.0;
  if (p < length - 1 && flag[p]) {
    const double a = in[idx[p]], b = in[idx[p + 1]];
    r = a < b ? b : a;
  } else if (p >= 1 && p - 1 < length - 1 && flag[p - 1]) {
    const double a = in[idx[p - 1]], b = in[idx[p]];
    r = a < b ? a : b;
  }
  out[p] = r;
}

extern "C" int sdm_sort_pair(sdm_ctx *ctx, double *out, int64_t n_out, const double *in,
                             const uint8_t *flag, const int64_t *idx, int64_t length) {
  ARG_TRY(ctx && n_out >= 0 && length >= 0);
  if (n_out == 0) return SDM_OK;
  ARG_TRY(out && in && flag && idx);
  hipLaunchKernelGGL(k_sort_pair, GRID1D(n_out), out, n_out, in, flag, idx, length);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- normalize (collisions_methods.py:633-662) -------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_norm_factor(const int64_t *__restrict__ cell_start, double *__restrict__ norm_factor,
              int64_t n_cell, double timestep, double dv) {
  const int64_t c = TID();
  if (c >= n_cell) return;
  const int64_t sd_num = cell_start[c + 1] - cell_start[c];
  norm_factor[c] = sd_num < 2 ? 0.0
                              : timestep / dv * (double)sd_num * (double)(sd_num - 1) / 2 /
                                    (double)(sd_num / 2);
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_normalize(double *__restrict__ prob, int64_t n_prob, const int64_t *__restrict__ cell_id,
            const int64_t *__restrict__ cell_idx, const double *__restrict__ norm_factor) {
  const int64_t d = TID();
  if (d >= n_prob) return;
  // reference quirk kept: the cell of RAW super-droplet #d, not of the pair's members
  prob[d] *= norm_factor[cell_idx[cell_id[d]]];
}

extern "C" int sdm_normalize(sdm_ctx *ctx, double *prob, int64_t n_prob, const int64_t *cell_id,
                             const int64_t *cell_idx, const int64_t *cell_start,
                             double *norm_factor, int64_t n_cell, double timestep, double dv) {
  ARG_TRY(ctx && n_prob >= 0 && n_cell >= 1 && cell_start && norm_factor);
  hipLaunchKernelGGL(k_norm_factor, GRID1D(n_cell), cell_start, norm_factor, n_cell, timestep,
                     dv);
  LAUNCH_CHECK();
  if (n_prob == 0) return SDM_OK;
  ARG_TRY(prob && cell_id && cell_idx);
  hipLaunchKernelGGL(k_normalize, GRID1D(n_prob), prob, n_prob, cell_id, cell_idx, norm_factor);
  LAUNCH_CHECK();
  return SDM_OK;
}

// pair_indices (collisions_methods.py:16-35)
__device__ __forceinline__ bool pair_indices(int64_t i, const int64_t *__restrict__ idx,
                                             const uint8_t *__restrict__ flag, double prob_like,
                                             int64_t &j, int64_t &k) {
  if (prob_like == 0) return true;
  const int64_t offset = 1 - (int64_t)flag[2 * i];
  j = idx[2 * i + offset];
  k = idx[2 * i + 1 + offset];
  return false;
}

// ---- scale_prob_for_adaptive_sdm_gamma (collisions_methods.py:330-405) -------------------
// serial reference: dt_todo[c] = min(dt_left[c], dt_max) folded with min over the cell's pairs of
// dt_optimal; min is order-independent, so a parallel (atomic, bit-pattern) min is exact.
__global__ void __launch_bounds__(SDM_BLOCK)
k_adaptive_init(double *__restrict__ dt_todo, double *__restrict__ cell_min,
                const double *__restrict__ dt_left, int64_t n_cell, double dt_max) {
  const int64_t c = TID();
  if (c >= n_cell) return;
  const double l = dt_left[c];
  dt_todo[c] = dt_max < l ? dt_max : l;  // Python min(l, dt_max)
  cell_min[c] = INFINITY;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_adaptive_min(const double *__restrict__ prob, const int64_t *__restrict__ idx, int64_t length,
               const int64_t *__restrict__ multiplicity, const int64_t *__restrict__ cell_id,
               double dt, double dt_min, const uint8_t *__restrict__ flag,
               double *__restrict__ cell_min) {
  const int64_t i = TID();
  bool active = i < length / 2;
  int64_t j = 0, k = 0;
  if (active) active = !pair_indices(i, idx, flag, prob[i], j, k);
  double dt_optimal = INFINITY;
  int64_t cid = -1;
  if (active) {
    const int64_t prop = multiplicity[j] / multiplicity[k];
    dt_optimal = dt * (double)prop / prob[i];
    dt_optimal = dt_min > dt_optimal ? dt_min : dt_optimal;  // Python max(): NaN dt_min = no clamp
    cid = cell_id[j];
  }
  // wave-aggregate when every active lane sits in the same cell (the common case)
  const unsigned long long am = __ballot(active);
  if (am == 0) return;
  const int first = __ffsll((long long)am) - 1;
  const int64_t cid0 = __shfl((long long)cid, first, 64);
  const bool uniform = __all(!active || cid == cid0);
  if (uniform) {
    const double m = wave_min_f64(dt_optimal);
    if (lane_id() == first) atomic_min_pos_f64(&cell_min[cid0], m);
  } else if (active) {
    atomic_min_pos_f64(&cell_min[cid], dt_optimal);
  }
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_adaptive_cells(double *__restrict__ dt_todo, const double *__restrict__ cell_min,
                 double *__restrict__ stats_dt_min, int64_t n_cell) {
  const int64_t c = TID();
  if (c >= n_cell) return;
  const double m = cell_min[c];
  if (m < dt_todo[c]) dt_todo[c] = m;
  // Python min(stats, m): NaN-sticky on the left operand
  const double s = stats_dt_min[c];
  stats_dt_min[c] = m < s ? m : s;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_adaptive_scale(double *__restrict__ prob, const int64_t *__restrict__ idx, int64_t length,
                 const int64_t *__restrict__ cell_id, double dt,
                 const uint8_t *__restrict__ flag, const double *__restrict__ dt_todo) {
  const int64_t i = TID();
  if (i >= length / 2) return;
  int64_t j, k;
  const double p = prob[i];
  if (pair_indices(i, idx, flag, p, j, k)) return;
  prob[i] = p * (dt_todo[cell_id[j]] / dt);
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_adaptive_finish(double *__restrict__ dt_left, const double *__restrict__ dt_todo,
                  int64_t *__restrict__ stats_n_substep, int64_t n_cell) {
  const int64_t c = TID();
  if (c >= n_cell) return;
  const double t = dt_todo[c];
  dt_left[c] -= t;
  if (t > 0) stats_n_substep[c] += 1;
}

extern "C" int sdm_scale_prob_for_adaptive_sdm_gamma(
    sdm_ctx *ctx, double *prob, const int64_t *idx, int64_t length, const int64_t *multiplicity,
    const int64_t *cell_id, double *dt_left, int64_t n_cell, double dt, double dt_min,
    double dt_max, const uint8_t *flag, int64_t *stats_n_substep, double *stats_dt_min) {
  ARG_TRY(ctx && length >= 0 && n_cell >= 1 && dt_left && stats_n_substep && stats_dt_min);
  ARG_TRY(!(dt_min <= 0));  // NaN = "no lower bound" (the reference's tests pass it)
  int rc = sdm_reserve(ctx, 2 * carve_size(sizeof(double) * n_cell));
  if (rc) return rc;
  Carver cv(ctx->arena);
  double *dt_todo = cv.take<double>(n_cell);
  double *cell_min = cv.take<double>(n_cell);
  hipLaunchKernelGGL(k_adaptive_init, GRID1D(n_cell), dt_todo, cell_min, dt_left, n_cell, dt_max);
  LAUNCH_CHECK();
  const int64_t n_pairs = length / 2;
  if (n_pairs > 0) {
    ARG_TRY(prob && idx && multiplicity && cell_id && flag);
    hipLaunchKernelGGL(k_adaptive_min, GRID1D(n_pairs), prob, idx, length, multiplicity, cell_id,
                       dt, dt_min, flag, cell_min);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_adaptive_cells, GRID1D(n_cell), dt_todo, cell_min, stats_dt_min, n_cell);
  LAUNCH_CHECK();
  if (n_pairs > 0) {
    hipLaunchKernelGGL(k_adaptive_scale, GRID1D(n_pairs), prob, idx, length, cell_id, dt, flag,
                       dt_todo);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_adaptive_finish, GRID1D(n_cell), dt_left, dt_todo, stats_n_substep,
                     n_cell);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- compute_gamma (collisions_methods.py:522-585) ---------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_compute_gamma(const double *prob, const double *__restrict__ rand,
                const int64_t *__restrict__ idx, int64_t length,
                const int64_t *__restrict__ multiplicity, const int64_t *__restrict__ cell_id,
                int64_t *__restrict__ collision_rate_deficit,
                int64_t *__restrict__ collision_rate, const uint8_t *__restrict__ flag,
                double *out) {
  const int64_t i = TID();
  const bool in_range = i < length / 2;
  double g = in_range ? ceil(prob[i] - rand[i]) : 0.0;
  int64_t j, k, cid = 0, done = 0, deficit = 0;
  if (in_range && !pair_indices(i, idx, flag, g, j, k)) {
    const int64_t nk = multiplicity[k];
    const int64_t prop = multiplicity[j] / nk;
    const int64_t gi = (int64_t)g;
    const int64_t gc = gi < prop ? gi : prop;
    cid = cell_id[j];
    done = gc * nk;
    deficit = (gi - gc) * nk;
    g = (double)gc;
  }
  wave_counter_add(collision_rate, cid, done, true);
  wave_counter_add(collision_rate_deficit, cid, deficit, true);
  if (in_range) out[i] = g;
}

extern "C" int sdm_compute_gamma(sdm_ctx *ctx, const double *prob, const double *rand,
                                 const int64_t *idx, int64_t length, const int64_t *multiplicity,
                                 const int64_t *cell_id, int64_t *collision_rate_deficit,
                                 int64_t *collision_rate, const uint8_t *flag, double *out) {
  ARG_TRY(ctx && length >= 0);
  if (length / 2 == 0) return SDM_OK;
  ARG_TRY(prob && rand && idx && multiplicity && cell_id && collision_rate_deficit &&
          collision_rate && flag && out);
  hipLaunchKernelGGL(k_compute_gamma, GRID1D(length / 2), prob, rand, idx, length, multiplicity,
                     cell_id, collision_rate_deficit, collision_rate, flag, out);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- adaptive_sdm_end (collisions_methods.py:313-328) ------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_adaptive_end(const double *__restrict__ dt_left, int64_t n_cell,
               const int64_t *__restrict__ cell_start, int64_t *__restrict__ out_last) {
  // out_last[0] = 1 + (largest i with dt_left[i] != 0), 0 if none (pre-zeroed)
  const int64_t i = TID();
  const bool nz = i < n_cell && dt_left[i] != 0;
  const unsigned long long m = __ballot(nz);
  if (m && lane_id() == 0) {
    const int64_t top = (i - lane_id()) + (63 - __clzll(m)) + 1;
    atomicMax((long long *)out_last, (long long)top);
  }
}

__global__ void k_adaptive_end_final(const int64_t *__restrict__ cell_start,
                                     const int64_t *__restrict__ last,
                                     int64_t *__restrict__ end) {
  const int64_t t = last[0];
  end[0] = t == 0 ? 0 : cell_start[t];
}

int sdm_adaptive_end_async(sdm_ctx *ctx, const double *dt_left, int64_t n_cell,
                           const int64_t *cell_start, int64_t *scratch2, int64_t *end_dev) {
  HIP_TRY(hipMemsetAsync(scratch2, 0, sizeof(int64_t), ctx->stream));
  hipLaunchKernelGGL(k_adaptive_end, GRID1D(n_cell), dt_left, n_cell, cell_start, scratch2);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_adaptive_end_final, dim3(1), dim3(1), 0, ctx->stream, cell_start,
                     scratch2, end_dev);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_adaptive_sdm_end(sdm_ctx *ctx, const double *dt_left, int64_t n_cell,
                                    const int64_t *cell_start, int64_t *end) {
  ARG_TRY(ctx && dt_left && cell_start && end && n_cell >= 1);
  int rc = sdm_adaptive_end_async(ctx, dt_left, n_cell, cell_start, ctx->dscal + 10,
                                  ctx->dscal + 11);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, ctx->dscal + 11, sizeof(int64_t), hipMemcpyDeviceToHost,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *end = ctx->mailbox[0];
  return SDM_OK;
}

// ---- collision_coalescence (collisions_methods.py:418-453) -------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_collision_coalescence(int64_t *__restrict__ multiplicity, const int64_t *__restrict__ idx,
                        int64_t length, double *__restrict__ attributes, int64_t n_attr,
                        int64_t n_sd, const double *__restrict__ gamma,
                        int64_t *__restrict__ healthy, const int64_t *__restrict__ cell_id,
                        int64_t *__restrict__ coalescence_rate,
                        const uint8_t *__restrict__ flag) {
  const int64_t i = TID();
  const double g = i < length / 2 ? gamma[i] : 0.0;
  int64_t j, k;
  const bool skip = i >= length / 2 || pair_indices(i, idx, flag, g, j, k);
  const int64_t nk = skip ? 0 : multiplicity[k];
  wave_counter_add(coalescence_rate, skip ? 0 : cell_id[j], (int64_t)(g * (double)nk), !skip);
  if (skip) return;
  coalesce_pair(j, k, g, multiplicity, attributes, n_attr, n_sd);
  if (multiplicity[k] == 0 || multiplicity[j] == 0) healthy[0] = 0;
}

extern "C" int sdm_collision_coalescence(sdm_ctx *ctx, int64_t *multiplicity, const int64_t *idx,
                                         int64_t length, double *attributes, int64_t n_attr,
                                         int64_t n_sd, const double *gamma, int64_t *healthy,
                                         const int64_t *cell_id, int64_t *coalescence_rate,
                                         const uint8_t *flag) {
  ARG_TRY(ctx && length >= 0 && n_attr >= 0 && n_sd >= 0);
  if (length / 2 == 0) return SDM_OK;
  ARG_TRY(multiplicity && idx && (attributes || n_attr == 0) && gamma && healthy && cell_id &&
          coalescence_rate && flag);
  hipLaunchKernelGGL(k_collision_coalescence, GRID1D(length / 2), multiplicity, idx, length,
                     attributes, n_attr, n_sd, gamma, healthy, cell_id, coalescence_rate, flag);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- collision_coalescence_breakup (collisions_methods.py:247-311) -----------------------
__device__ __forceinline__ void add_i64(int64_t *p, int64_t v) {
  if (v != 0) atomicAdd((unsigned long long *)p, (unsigned long long)v);
}

// one pair: break_up :135-175 ; returns overflow
__device__ bool break_up_pair(int64_t j, int64_t k, int64_t cid, double gamma,
                              int64_t *multiplicity, double *attributes, int64_t n_attr,
                              int64_t n_sd, double fragment_mass_i, int64_t max_multiplicity,
                              int64_t *breakup_rate, int64_t *breakup_rate_deficit,
                              const double *particle_mass) {
  double take_from_j, new_mult_k;
  int64_t gamma_j_k;
  bool overflow;
  const int64_t nk = multiplicity[k];
  compute_transfer_multiplicities(gamma, multiplicity[j], nk, particle_mass[j], particle_mass[k],
                                  fragment_mass_i, max_multiplicity, take_from_j, new_mult_k,
                                  gamma_j_k, overflow);
  const double gamma_deficit = gamma - (double)gamma_j_k;
  add_i64(&breakup_rate[cid], gamma_j_k * nk);
  add_i64(&breakup_rate_deficit[cid], (int64_t)(gamma_deficit * (double)nk));
  apply_breakup_transfer(j, k, take_from_j, new_mult_k, multiplicity, attributes, n_attr, n_sd);
  return overflow;
}

// break_up_while :178-243
__device__ bool break_up_while_pair(int64_t j, int64_t k, int64_t cid, double gamma,
                                    int64_t *multiplicity, double *attributes, int64_t n_attr,
                                    int64_t n_sd, double fragment_mass_i,
                                    int64_t max_multiplicity, int64_t *breakup_rate,
                                    int64_t *breakup_rate_deficit,
                                    const double *particle_mass) {
  double gamma_deficit = gamma;
  bool overflow = false;
  while (gamma_deficit > 0) {
    double take_from_j, new_mult_k, gamma_j_k;
    const int64_t nj = multiplicity[j], nk = multiplicity[k];
    if (nk == nj) {
      take_from_j = (double)nj;
      new_mult_k = (particle_mass[j] + particle_mass[k]) / fragment_mass_i * (double)nk;
      if (new_mult_k > (double)max_multiplicity) {
        add_i64(&breakup_rate_deficit[cid], (int64_t)(gamma_deficit * (double)nk));
        overflow = true;
        break;
      }
      gamma_j_k = gamma_deficit;
    } else {
      if (nk > nj) { const int64_t t = j; j = k; k = t; }
      int64_t g_int;
      compute_transfer_multiplicities(gamma_deficit, multiplicity[j], multiplicity[k],
                                      particle_mass[j], particle_mass[k], fragment_mass_i,
                                      max_multiplicity, take_from_j, new_mult_k, g_int,
                                      overflow);
      gamma_j_k = (double)g_int;
      if (g_int == 0) break;  // (safety deviation: fused.hip, resolve_collision)
    }
    add_i64(&breakup_rate[cid], (int64_t)(gamma_j_k * (double)multiplicity[k]));
    gamma_deficit -= gamma_j_k;
    apply_breakup_transfer(j, k, take_from_j, new_mult_k, multiplicity, attributes, n_attr,
                           n_sd);
  }
  add_i64(&breakup_rate_deficit[cid], (int64_t)(gamma_deficit * (double)multiplicity[k]));
  return overflow;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_collision_coalescence_breakup(int64_t *multiplicity, const int64_t *__restrict__ idx,
                                int64_t length, double *attributes, int64_t n_attr,
                                int64_t n_sd, const double *__restrict__ gamma,
                                const double *__restrict__ rand, const double *__restrict__ Ec,
                                const double *__restrict__ Eb,
                                const double *__restrict__ fragment_mass,
                                int64_t *__restrict__ healthy,
                                const int64_t *__restrict__ cell_id, int64_t *coalescence_rate,
                                int64_t *breakup_rate, int64_t *breakup_rate_deficit,
                                const uint8_t *__restrict__ flag, int64_t max_multiplicity,
                                const double *particle_mass, int handle_all_breakups,
                                int64_t *n_overflow) {
  const int64_t i = TID();
  if (i >= length / 2) return;
  const double g = gamma[i];
  int64_t j, k;
  if (pair_indices(i, idx, flag, g, j, k)) return;
  const double r = rand[i], ec = Ec[i], eb = Eb[i];
  if (r - (ec + (1 - ec) * eb) > 0) return;  // bounce
  const int64_t cid = cell_id[j];
  if (r - ec < 0) {
    add_i64(&coalescence_rate[cid], (int64_t)(g * (double)multiplicity[k]));
    coalesce_pair(j, k, g, multiplicity, attributes, n_attr, n_sd);
  } else {
    const bool ovf =
        handle_all_breakups
            ? break_up_while_pair(j, k, cid, g, multiplicity, attributes, n_attr, n_sd,
                                  fragment_mass[i], max_multiplicity, breakup_rate,
                                  breakup_rate_deficit, particle_mass)
            : break_up_pair(j, k, cid, g, multiplicity, attributes, n_attr, n_sd,
                            fragment_mass[i], max_multiplicity, breakup_rate,
                            breakup_rate_deficit, particle_mass);
    if (ovf && n_overflow) add_i64(n_overflow, 1);
  }
  if (multiplicity[k] == 0 || multiplicity[j] == 0) healthy[0] = 0;
}

extern "C" int sdm_collision_coalescence_breakup(
    sdm_ctx *ctx, int64_t *multiplicity, const int64_t *idx, int64_t length, double *attributes,
    int64_t n_attr, int64_t n_sd, const double *gamma, const double *rand, const double *Ec,
    const double *Eb, const double *fragment_mass, int64_t *healthy, const int64_t *cell_id,
    int64_t *coalescence_rate, int64_t *breakup_rate, int64_t *breakup_rate_deficit,
    const uint8_t *flag, int64_t max_multiplicity, const double *particle_mass,
    int handle_all_breakups, int64_t *n_overflow) {
  ARG_TRY(ctx && length >= 0 && n_attr >= 0);
  if (length / 2 == 0) return SDM_OK;
  ARG_TRY(multiplicity && idx && attributes && gamma && rand && Ec && Eb && fragment_mass &&
          healthy && cell_id && coalescence_rate && breakup_rate && breakup_rate_deficit &&
          flag && particle_mass);
  hipLaunchKernelGGL(k_collision_coalescence_breakup, GRID1D(length / 2), multiplicity, idx,
                     length, attributes, n_attr, n_sd, gamma, rand, Ec, Eb, fragment_mass,
                     healthy, cell_id, coalescence_rate, breakup_rate, breakup_rate_deficit,
                     flag, max_multiplicity, particle_mass, handle_all_breakups, n_overflow);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- linear_collection_efficiency (collisions_methods.py:743-782) ------------------------
struct BerryParams { double p[13]; };

__global__ void __launch_bounds__(SDM_BLOCK)
k_linear_collection_efficiency(BerryParams P, double *__restrict__ out, int64_t n_out,
                               const double *__restrict__ radii,
                               const uint8_t *__restrict__ flag,
                               const int64_t *__restrict__ idx, int64_t length, double unit) {
  const int64_t d = TID();
  if (d >= n_out) return;
  double r = 0.0;
  int64_t i = -1;
  if (2 * d < length - 1 && flag[2 * d]) i = 2 * d;
  else if (2 * d + 1 < length - 1 && flag[2 * d + 1]) i = 2 * d + 1;
  if (i >= 0) r = linear_collection_efficiency(P.p, radii[idx[i]], radii[idx[i + 1]], unit);
  out[d] = r;
}

extern "C" int sdm_linear_collection_efficiency(sdm_ctx *ctx, const double params[13],
                                                double *output, int64_t n_out,
                                                const double *radii, const uint8_t *flag,
                                                const int64_t *idx, int64_t length,
                                                double unit) {
  ARG_TRY(ctx && params && n_out >= 0 && length >= 0);
  if (n_out == 0) return SDM_OK;
  ARG_TRY(output && radii && flag && idx);
  BerryParams P;
  memcpy(P.p, params, sizeof(P.p));
  hipLaunchKernelGGL(k_linear_collection_efficiency, GRID1D(n_out), P, output, n_out, radii,
                     flag, idx, length, unit);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- interpolation / volume / mass ---------------------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_interpolation(double *__restrict__ out, const double *__restrict__ radius, int64_t n,
                double factor, const double *__restrict__ b, const double *__restrict__ c,
                int64_t table_len) {
  const int64_t i = TID();
  if (i < n) out[i] = gk_interpolate(radius[i], factor, b, c, table_len);
}

extern "C" int sdm_interpolation(sdm_ctx *ctx, double *output, const double *radius, int64_t n,
                                 double factor, const double *b, const double *c,
                                 int64_t table_len) {
  ARG_TRY(ctx && n >= 0 && table_len >= 1);
  if (n == 0) return SDM_OK;
  ARG_TRY(output && radius && b && c);
  hipLaunchKernelGGL(k_interpolation, GRID1D(n), output, radius, n, factor, b, c, table_len);
  LAUNCH_CHECK();
  return SDM_OK;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_scale(double *__restrict__ out, const double *__restrict__ in, int64_t n, double s, int div) {
  const int64_t i = TID();
  if (i < n) out[i] = div ? in[i] / s : s * in[i];
}

extern "C" int sdm_volume_of_water_mass(sdm_ctx *ctx, double *volume, const double *mass,
                                        int64_t n, double rho_w) {
  ARG_TRY(ctx && n >= 0 && (n == 0 || (volume && mass)));
  if (n == 0) return SDM_OK;
  hipLaunchKernelGGL(k_scale, GRID1D(n), volume, mass, n, rho_w, 1);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_mass_of_water_volume(sdm_ctx *ctx, double *mass, const double *volume,
                                        int64_t n, double rho_w) {
  ARG_TRY(ctx && n >= 0 && (n == 0 || (volume && mass)));
  if (n == 0) return SDM_OK;
  hipLaunchKernelGGL(k_scale, GRID1D(n), mass, volume, n, rho_w, 0);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- fragmentation ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_exp_fragmentation(double *__restrict__ n_fragment, double scale,
                    double *__restrict__ frag_volume, const double *__restrict__ x_plus_y,
                    const double *__restrict__ rand, int64_t n, double vmin, double nfmax,
                    double tol) {
  const int64_t i = TID();
  if (i >= n) return;
  const double a = 1 - rand[i];
  double fv = -scale * sdm_log(a > tol ? a : tol);
  double nf;
  fragmentation_limiters(nf, fv, vmin, nfmax, x_plus_y[i]);
  frag_volume[i] = fv;
  n_fragment[i] = nf;
}

extern "C" int sdm_exp_fragmentation(sdm_ctx *ctx, double *n_fragment, double scale,
                                     double *frag_volume, const double *x_plus_y,
                                     const double *rand, int64_t n, double vmin, double nfmax,
                                     double tol) {
  ARG_TRY(ctx && n >= 0);
  if (n == 0) return SDM_OK;
  ARG_TRY(n_fragment && frag_volume && x_plus_y && rand);
  hipLaunchKernelGGL(k_exp_fragmentation, GRID1D(n), n_fragment, scale, frag_volume, x_plus_y,
                     rand, n, vmin, nfmax, tol);
  LAUNCH_CHECK();
  return SDM_OK;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_gauss_fragmentation(double *__restrict__ n_fragment, double mu, double sigma,
                      double *__restrict__ frag_volume, const double *__restrict__ x_plus_y,
                      const double *__restrict__ rand, int64_t n, double vmin, double nfmax,
                      double VA, double Vb) {
  const int64_t i = TID();
  if (i >= n) return;
  double fv = mu + sigma * erfinv_approx(rand[i], VA, Vb), nf;
  fragmentation_limiters(nf, fv, vmin, nfmax, x_plus_y[i]);
  frag_volume[i] = fv;
  n_fragment[i] = nf;
}

extern "C" int sdm_gauss_fragmentation(sdm_ctx *ctx, double *n_fragment, double mu, double sigma,
                                       double *frag_volume, const double *x_plus_y,
                                       const double *rand, int64_t n, double vmin, double nfmax,
                                       const double consts[2]) {
  ARG_TRY(ctx && n >= 0 && consts);
  if (n == 0) return SDM_OK;
  ARG_TRY(n_fragment && frag_volume && x_plus_y && rand);
  hipLaunchKernelGGL(k_gauss_fragmentation, GRID1D(n), n_fragment, mu, sigma, frag_volume,
                     x_plus_y, rand, n, vmin, nfmax, consts[0], consts[1]);
  LAUNCH_CHECK();
  return SDM_OK;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_feingold1988_fragmentation(double *__restrict__ n_fragment, double scale,
                             double *__restrict__ frag_volume,
                             const double *__restrict__ x_plus_y,
                             const double *__restrict__ rand, int64_t n, double fragtol,
                             double vmin, double nfmax) {
  const int64_t i = TID();
  if (i >= n) return;
  const double a = 1 - rand[i] * scale / x_plus_y[i];
  double fv = -scale * sdm_log(a > fragtol ? a : fragtol), nf;
  fragmentation_limiters(nf, fv, vmin, nfmax, x_plus_y[i]);
  frag_volume[i] = fv;
  n_fragment[i] = nf;
}

extern "C" int sdm_feingold1988_fragmentation(sdm_ctx *ctx, double *n_fragment, double scale,
                                              double *frag_volume, const double *x_plus_y,
                                              const double *rand, int64_t n, double fragtol,
                                              double vmin, double nfmax) {
  ARG_TRY(ctx && n >= 0);
  if (n == 0) return SDM_OK;
  ARG_TRY(n_fragment && frag_volume && x_plus_y && rand);
  hipLaunchKernelGGL(k_feingold1988_fragmentation, GRID1D(n), n_fragment, scale, frag_volume,
                     x_plus_y, rand, n, fragtol, vmin, nfmax);
  LAUNCH_CHECK();
  return SDM_OK;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_slams_fragmentation(double *__restrict__ n_fragment, double *__restrict__ frag_volume,
                      const double *__restrict__ x_plus_y, double *__restrict__ probs,
                      const double *__restrict__ rand, int64_t n, double vmin, double nfmax) {
  const int64_t i = TID();
  if (i >= n) return;
  double p = 0.0, nf = 1;
  for (int k = 0; k < 22; ++k) {
    p += 0.91 * sdm_pow((double)(k + 2), -1.56);
    if (rand[i] < p) { nf = k + 2; break; }
  }
  probs[i] = p;
  double fv = x_plus_y[i] / nf;
  fragmentation_limiters(nf, fv, vmin, nfmax, x_plus_y[i]);
  frag_volume[i] = fv;
  n_fragment[i] = nf;
}

extern "C" int sdm_slams_fragmentation(sdm_ctx *ctx, double *n_fragment, double *frag_volume,
                                       const double *x_plus_y, double *probs, const double *rand,
                                       int64_t n, double vmin, double nfmax) {
  ARG_TRY(ctx && n >= 0);
  if (n == 0) return SDM_OK;
  ARG_TRY(n_fragment && frag_volume && x_plus_y && probs && rand);
  hipLaunchKernelGGL(k_slams_fragmentation, GRID1D(n), n_fragment, frag_volume, x_plus_y, probs,
                     rand, n, vmin, nfmax);
  LAUNCH_CHECK();
  return SDM_OK;
}

struct LL82Consts { double k[4]; };

__global__ void __launch_bounds__(SDM_BLOCK)
k_ll82_fragmentation(double *__restrict__ n_fragment, const double *__restrict__ CKE,
                     const double *__restrict__ W, const double *__restrict__ W2,
                     const double *__restrict__ St, const double *__restrict__ ds,
                     const double *__restrict__ dl, const double *__restrict__ dcoal,
                     double *__restrict__ frag_volume, const double *__restrict__ x_plus_y,
                     double *__restrict__ rand, int64_t n, double vmin, double nfmax,
                     double *__restrict__ Rf, double *__restrict__ Rs, double *__restrict__ Rd,
                     double tol, LL82Consts K) {
  const int64_t i = TID();
  if (i >= n) return;
  double r = rand[i], rf = Rf[i], rs = Rs[i], rd = Rd[i];
  double fv = ll82_fragment_volume(CKE[i], W[i], W2[i], St[i], ds[i], dl[i], dcoal[i], &r, &rf,
                                   &rs, &rd, tol, K.k);
  rand[i] = r; Rf[i] = rf; Rs[i] = rs; Rd[i] = rd;
  double nf;
  fragmentation_limiters(nf, fv, vmin, nfmax, x_plus_y[i]);
  frag_volume[i] = fv;
  n_fragment[i] = nf;
}

extern "C" int sdm_ll82_fragmentation(sdm_ctx *ctx, double *n_fragment, const double *CKE,
                                      const double *W, const double *W2, const double *St,
                                      const double *ds, const double *dl, const double *dcoal,
                                      double *frag_volume, const double *x_plus_y, double *rand,
                                      int64_t n, double vmin, double nfmax, double *Rf,
                                      double *Rs, double *Rd, double tol,
                                      const double consts[4]) {
  ARG_TRY(ctx && n >= 0 && consts);
  if (n == 0) return SDM_OK;
  ARG_TRY(n_fragment && CKE && W && W2 && St && ds && dl && dcoal && frag_volume && x_plus_y &&
          rand && Rf && Rs && Rd);
  LL82Consts K;
  memcpy(K.k, consts, sizeof(K.k));
  hipLaunchKernelGGL(k_ll82_fragmentation, GRID1D(n), n_fragment, CKE, W, W2, St, ds, dl, dcoal,
                     frag_volume, x_plus_y, rand, n, vmin, nfmax, Rf, Rs, Rd, tol, K);
  LAUNCH_CHECK();
  return SDM_OK;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_ll82_coalescence_check(double *__restrict__ Ec, const double *__restrict__ dl, int64_t n) {
  const int64_t i = TID();
  if (i < n && dl[i] < 0.4e-3) Ec[i] = 1.0;
}

extern "C" int sdm_ll82_coalescence_check(sdm_ctx *ctx, double *Ec, const double *dl,
                                          int64_t n) {
  ARG_TRY(ctx && n >= 0);
  if (n == 0) return SDM_OK;
  ARG_TRY(Ec && dl);
  hipLaunchKernelGGL(k_ll82_coalescence_check, GRID1D(n), Ec, dl, n);
  LAUNCH_CHECK();
  return SDM_OK;
}

struct StraubConsts { double k[6]; };

__global__ void __launch_bounds__(SDM_BLOCK)
k_straub_fragmentation(double *__restrict__ n_fragment, const double *__restrict__ CW,
                       const double *__restrict__ gam, const double *__restrict__ ds,
                       double *__restrict__ frag_volume, const double *__restrict__ v_max,
                       const double *__restrict__ x_plus_y, const double *__restrict__ rand,
                       int64_t n, double vmin, double nfmax, double *Nr1, double *Nr2,
                       double *Nr3, double *Nr4, double *Nrt, double *d34, StraubConsts K) {
  const int64_t i = TID();
  if (i >= n) return;
  StraubTmp T = {Nr1[i], Nr2[i], Nr3[i], Nr4[i], Nrt[i], d34[i]};
  double fv = straub_fragment_volume(CW[i], gam[i], ds[i], v_max[i], rand[i], K.k, T);
  Nr1[i] = T.Nr1; Nr2[i] = T.Nr2; Nr3[i] = T.Nr3; Nr4[i] = T.Nr4; Nrt[i] = T.Nrt; d34[i] = T.d34;
  double nf;
  fragmentation_limiters(nf, fv, vmin, nfmax, x_plus_y[i]);
  frag_volume[i] = fv;
  n_fragment[i] = nf;
}

extern "C" int sdm_straub_fragmentation(sdm_ctx *ctx, double *n_fragment, const double *CW,
                                        const double *gam, const double *ds, double *frag_volume,
                                        const double *v_max, const double *x_plus_y,
                                        const double *rand, int64_t n, double vmin, double nfmax,
                                        double *Nr1, double *Nr2, double *Nr3, double *Nr4,
                                        double *Nrt, double *d34, const double consts[6]) {
  ARG_TRY(ctx && n >= 0 && consts);
  if (n == 0) return SDM_OK;
  ARG_TRY(n_fragment && CW && gam && ds && frag_volume && v_max && x_plus_y && rand && Nr1 &&
          Nr2 && Nr3 && Nr4 && Nrt && d34);
  StraubConsts K;
  memcpy(K.k, consts, sizeof(K.k));
  hipLaunchKernelGGL(k_straub_fragmentation, GRID1D(n), n_fragment, CW, gam, ds, frag_volume,
                     v_max, x_plus_y, rand, n, vmin, nfmax, Nr1, Nr2, Nr3, Nr4, Nrt, d34, K);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- terminal velocities besides the Gunn-Kinzer table (terminal_velocity_methods.py:32-66) ---
struct TermVelConsts { double k[5]; };
struct PowerSeriesTerms { double prefactor[16], power[16]; };

__global__ void __launch_bounds__(SDM_BLOCK)
k_terminal_velocity(double *__restrict__ values, const double *__restrict__ radius, int64_t n,
                    TermVelConsts K) {
  const int64_t i = TID();
  if (i >= n) return;
  const double r = radius[i];
  values[i] = r < K.k[3] ? K.k[0] * (r * r) : (r < K.k[4] ? K.k[1] * r : K.k[2] * sdm_pow(r, 0.5));
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_power_series(double *__restrict__ values, const double *__restrict__ radius, int64_t n,
               int num_terms, PowerSeriesTerms T) {
  const int64_t i = TID();
  if (i >= n) return;
  double v = 0.0;
  for (int j = 0; j < num_terms; ++j) v = v + T.prefactor[j] * sdm_pow(radius[i], T.power[j] * 3);
  values[i] = v;
}

extern "C" int sdm_terminal_velocity(sdm_ctx *ctx, double *values, const double *radius,
                                     int64_t n, const double consts[5]) {
  ARG_TRY(ctx && n >= 0 && consts);
  if (n == 0) return SDM_OK;
  ARG_TRY(values && radius);
  TermVelConsts K;
  memcpy(K.k, consts, sizeof(K.k));
  hipLaunchKernelGGL(k_terminal_velocity, GRID1D(n), values, radius, n, K);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_power_series(sdm_ctx *ctx, double *values, const double *radius, int64_t n,
                                int num_terms, const double *prefactors, const double *powers) {
  ARG_TRY(ctx && n >= 0 && num_terms >= 0 && num_terms <= 16);
  ARG_TRY(num_terms == 0 || (prefactors && powers));
  if (n == 0) return SDM_OK;
  ARG_TRY(values && radius);
  PowerSeriesTerms T;
  memset(&T, 0, sizeof(T));
  for (int j = 0; j < num_terms; ++j) { T.prefactor[j] = prefactors[j]; T.power[j] = powers[j]; }
  hipLaunchKernelGGL(k_power_series, GRID1D(n), values, radius, n, num_terms, T);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- moments (moments_methods.py:14-99) ----------------------------------------------------
// After a collision step the state is sorted by cell (or is one cell), so a wave's 64 SDs nearly
// always share their cell: the wave then folds its terms with shuffles and issues one atomic per
// (wave, rank) instead of 64 to the same address.  Mixed waves fall back to per-lane atomics.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Every wave walks a contiguous chunk of positions and keeps running sums for "its current
// cell" in registers, flushing them with one atomic per (cell change, rank): a single-cell state
// of 2^20 super-droplets costs 2 k atomics per rank instead of 16 k on one address (which take
// ~60 ns each there).  Up to MOM_MAXR ranks per pass.
#define MOM_MAXR 4
#define MOM_GRID 512

__global__ void __launch_bounds__(SDM_BLOCK)
k_moments(double *__restrict__ moment_0, double *__restrict__ moments,
          const int64_t *__restrict__ multiplicity, const double *__restrict__ attr_data,
          const int64_t *__restrict__ cell_id, const int64_t *__restrict__ idx, int64_t length,
          const double *__restrict__ ranks, int rank_first, int n_pass, int with_m0,
          int64_t n_cell, double min_x, double max_x, const double *__restrict__ x_attr,
          const double *__restrict__ weighting_attribute, double weighting_rank) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * (SDM_BLOCK / SDM_WAVE);
  const int64_t wave = (int64_t)blockIdx.x * (SDM_BLOCK / SDM_WAVE) + threadIdx.x / SDM_WAVE;
  const int64_t chunk = (((length + n_waves - 1) / n_waves) + 63) & ~(int64_t)63;
  const int64_t begin = wave * chunk;
  const int64_t end = begin + chunk < length ? begin + chunk : length;
  int cur = -1;  // cell of the running sums
  double a0 = 0.0, acc[MOM_MAXR] = {0.0, 0.0, 0.0, 0.0};
  auto flush = [&]() {
    if (cur >= 0 && lane == 0) {
      if (with_m0) atomicAdd(&moment_0[cur], a0);
      for (int k = 0; k < n_pass; ++k)
        atomicAdd(&moments[(int64_t)(rank_first + k) * n_cell + cur], acc[k]);
    }
    a0 = 0.0;
    for (int k = 0; k < MOM_MAXR; ++k) acc[k] = 0.0;
  };
  for (int64_t t0 = begin; t0 < end; t0 += 64) {
    const int64_t t = t0 + lane;
    bool live = t < end;
    int64_t i = 0;
    if (live) {
      i = idx[t];
      const double x = x_attr[i];
      live = min_x <= x && x < max_x;
    }
    const int c = live ? (int)cell_id[i] : -1;
    int lo = live ? c : 0x7fffffff, hi = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o, 64));
      hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if (hi < 0) continue;  // nothing in range in this round
    const double w = !live ? 0.0 : (double)multiplicity[i] *
                     (weighting_rank == 0 ? 1.0 : sdm_pow(weighting_attribute[i], weighting_rank));
    double term[MOM_MAXR];
    for (int k = 0; k < MOM_MAXR; ++k)
      term[k] = (live && k < n_pass) ? w * sdm_pow(attr_data[i], ranks[rank_first + k]) : 0.0;
    if (lo == hi) {  // the usual case after a collision step: sorted by cell, or one cell
      if (hi != cur) {
        flush();
        cur = hi;
      }
      a0 += wave_sum_f64(w);
      for (int k = 0; k < MOM_MAXR; ++k)
        if (k < n_pass) acc[k] += wave_sum_f64(term[k]);
    } else if (live) {
      if (with_m0) atomicAdd(&moment_0[c], w);
      for (int k = 0; k < n_pass; ++k)
        atomicAdd(&moments[(int64_t)(rank_first + k) * n_cell + c], term[k]);
    }
  }
  flush();
}

// ---- spectrum_moments (moments_methods.py:100-147) -----------------------------------------
// bin edges staged in LDS; the first matching bin wins exactly as in the reference's scan
__global__ void __launch_bounds__(SDM_BLOCK)
k_spectrum_moments(double *__restrict__ moment_0, double *__restrict__ moments,
                   const int64_t *__restrict__ multiplicity,
                   const double *__restrict__ attr_data, const int64_t *__restrict__ cell_id,
                   const int64_t *__restrict__ idx, int64_t length, double rank,
                   const double *__restrict__ x_bins, int n_bins, int64_t n_cell,
                   const double *__restrict__ x_attr,
                   const double *__restrict__ weighting_attribute, double weighting_rank) {
  extern __shared__ double edges[];
  for (int k = threadIdx.x; k <= n_bins; k += blockDim.x) edges[k] = x_bins[k];
  __syncthreads();
  const int64_t t = TID();
  if (t >= length) return;
  const int64_t i = idx[t];
  const double x = x_attr[i];
  int bin = -1;
  for (int k = 0; k < n_bins; ++k)
    if (edges[k] <= x && x < edges[k + 1]) { bin = k; break; }
  if (bin < 0) return;
  const double w = (double)multiplicity[i] * sdm_pow(weighting_attribute[i], weighting_rank);
  const int64_t at = bin * n_cell + cell_id[i];
  atomicAdd(&moment_0[at], w);
  atomicAdd(&moments[at], w * sdm_pow(attr_data[i], rank));
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_spectrum_divide(const double *__restrict__ moment_0, double *__restrict__ moments, int64_t n) {
  const int64_t t = TID();
  if (t >= n) return;
  const double m0 = moment_0[t];
  moments[t] = m0 != 0 ? moments[t] / m0 : 0.0;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_moments_divide(const double *__restrict__ moment_0, double *__restrict__ moments,
                 int64_t n_ranks, int64_t n_cell) {
  const int64_t t = TID();
  if (t >= n_ranks * n_cell) return;
  const double m0 = moment_0[t % n_cell];
  moments[t] = m0 != 0 ? moments[t] / m0 : 0.0;
}

extern "C" int sdm_moments(sdm_ctx *ctx, double *moment_0, double *moments,
                           const int64_t *multiplicity, const double *attr_data,
                           const int64_t *cell_id, const int64_t *idx, int64_t length,
                           const double *ranks, int64_t n_ranks, int64_t n_cell, double min_x,
                           double max_x, const double *x_attr,
                           const double *weighting_attribute, double weighting_rank,
                           int skip_division_by_m0) {
  ARG_TRY(ctx && moment_0 && n_cell >= 1 && n_ranks >= 0 && length >= 0);
  ARG_TRY(n_ranks == 0 || (moments && ranks && attr_data));
  HIP_TRY(hipMemsetAsync(moment_0, 0, sizeof(double) * n_cell, ctx->stream));
  if (n_ranks > 0)
    HIP_TRY(hipMemsetAsync(moments, 0, sizeof(double) * n_ranks * n_cell, ctx->stream));
  if (length > 0) {
    ARG_TRY(multiplicity && cell_id && idx && x_attr && weighting_attribute);
    const unsigned grid = grid_for(length) < MOM_GRID ? grid_for(length) : MOM_GRID;
    for (int first = 0; first == 0 || first < n_ranks; first += MOM_MAXR) {
      const int n_pass = (int)(n_ranks - first < MOM_MAXR ? n_ranks - first : MOM_MAXR);
      hipLaunchKernelGGL(k_moments, dim3(grid), dim3(SDM_BLOCK), 0, ctx->stream, moment_0,
                         moments, multiplicity, attr_data, cell_id, idx, length, ranks, first,
                         n_pass, first == 0 ? 1 : 0, n_cell, min_x, max_x, x_attr,
                         weighting_attribute, weighting_rank);
      LAUNCH_CHECK();
    }
  }
  if (!skip_division_by_m0 && n_ranks > 0) {
    hipLaunchKernelGGL(k_moments_divide, GRID1D(n_ranks * n_cell), moment_0, moments, n_ranks,
                       n_cell);
    LAUNCH_CHECK();
  }
  return SDM_OK;
}

extern "C" int sdm_spectrum_moments(sdm_ctx *ctx, double *moment_0, double *moments,
                                    const int64_t *multiplicity, const double *attr_data,
                                    const int64_t *cell_id, const int64_t *idx, int64_t length,
                                    double rank, const double *x_bins, int64_t n_bins,
                                    int64_t n_cell, const double *x_attr,
                                    const double *weighting_attribute, double weighting_rank) {
  ARG_TRY(ctx && moment_0 && moments && x_bins && n_cell >= 1 && length >= 0);
  ARG_TRY(n_bins >= 1 && n_bins < 8000);  // the edges live in LDS
  HIP_TRY(hipMemsetAsync(moment_0, 0, sizeof(double) * n_bins * n_cell, ctx->stream));
  HIP_TRY(hipMemsetAsync(moments, 0, sizeof(double) * n_bins * n_cell, ctx->stream));
  if (length > 0) {
    ARG_TRY(multiplicity && attr_data && cell_id && idx && x_attr && weighting_attribute);
    hipLaunchKernelGGL(k_spectrum_moments, dim3((unsigned)((length + SDM_BLOCK - 1) / SDM_BLOCK)),
                       dim3(SDM_BLOCK), sizeof(double) * (n_bins + 1), ctx->stream, moment_0,
                       moments, multiplicity, attr_data, cell_id, idx, length, rank, x_bins,
                       (int)n_bins, n_cell, x_attr, weighting_attribute, weighting_rank);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_spectrum_divide, GRID1D(n_bins * n_cell), moment_0, moments,
                     n_bins * n_cell);
  LAUNCH_CHECK();
  return SDM_OK;
}
