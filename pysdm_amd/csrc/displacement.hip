// Particle displacement (advection + sedimentation) -- the step that precedes the collision
// path in multi-dimensional set-ups and unsorts the state.  Device counterparts of
// PySDM/backends/impl_numba/methods/displacement_methods.py; see include/sdm_hip.h.
#include "common.h"

#include <cstring>

#define GRID1D(n) dim3(grid_for(n)), dim3(SDM_BLOCK), 0, ctx->stream
#define TID() ((int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x)

struct CourantShape { int64_t s[3]; };

// displacement_methods.py:14-129 with physics/particle_advection/{implicit,explicit}_in_space.py
__global__ void __launch_bounds__(SDM_BLOCK)
k_calculate_displacement(int dim, int n_dims, int scheme, double *__restrict__ displacement,
                         const double *__restrict__ courant, CourantShape shape,
                         const int64_t *__restrict__ cell_origin,
                         const double *__restrict__ position_in_cell, int64_t n_sd,
                         double n_substeps) {
  const int64_t droplet = TID();
  if (droplet >= n_sd) return;
  int64_t l = 0, r = 0;
  for (int d = 0; d < n_dims; ++d) {
    const int64_t o = cell_origin[d * n_sd + droplet];
    l = l * shape.s[d] + o;
    r = r * shape.s[d] + o + (d == dim);
  }
  const double x = position_in_cell[dim * n_sd + droplet];
  const double c_l = courant[l] / n_substeps, c_r = courant[r] / n_substeps;
  double v = c_l * (1 - x) + c_r * x;
  if (scheme == 0) v = v / (1 - c_r + c_l);
  displacement[dim * n_sd + droplet] = v;
}

extern "C" int sdm_calculate_displacement(sdm_ctx *ctx, int dim, int n_dims, int scheme,
                                          double *displacement, const double *courant,
                                          const int64_t *courant_shape,
                                          const int64_t *cell_origin,
                                          const double *position_in_cell, int64_t n_sd,
                                          double n_substeps) {
  ARG_TRY(ctx && n_dims >= 1 && n_dims <= 3 && dim >= 0 && dim < n_dims && n_sd >= 0);
  ARG_TRY(scheme == 0 || scheme == 1);
  if (n_sd == 0) return SDM_OK;
  ARG_TRY(displacement && courant && courant_shape && cell_origin && position_in_cell);
  CourantShape shape = {{1, 1, 1}};
  for (int d = 0; d < n_dims; ++d) {
    ARG_TRY(courant_shape[d] >= 1);
    shape.s[d] = courant_shape[d];
  }
  hipLaunchKernelGGL(k_calculate_displacement, GRID1D(n_sd), dim, n_dims, scheme, displacement,
                     courant, shape, cell_origin, position_in_cell, n_sd, n_substeps);
  LAUNCH_CHECK();
  return SDM_OK;
}

// displacement_methods.py:131-166: one partial sum per block (fixed order), folded by one thread
__global__ void __launch_bounds__(SDM_BLOCK)
k_flag_precipitated(const int64_t *__restrict__ cell_origin,
                    const double *__restrict__ position_in_cell,
                    const double *__restrict__ water_mass,
                    const int64_t *__restrict__ multiplicity, int64_t *__restrict__ idx,
                    int64_t length, int64_t n_sd, int n_dims, int64_t *__restrict__ healthy,
                    double level, const double *__restrict__ displacement,
                    double *__restrict__ partial) {
  __shared__ double sm[SDM_BLOCK / SDM_WAVE];
  const int64_t i = TID();
  const int64_t last = (int64_t)(n_dims - 1) * n_sd;
  double mass = 0.0;
  if (i < length) {
    const int64_t k = idx[i];
    const double z = (double)cell_origin[last + k] + position_in_cell[last + k];
    if (displacement[last + k] < 0 && z < level) {
      mass = fabs(water_mass[k]) * (double)multiplicity[k];
      idx[i] = n_sd;
      healthy[0] = 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mass += __shfl_xor(mass, o, 64);
  if (lane_id() == 0) sm[threadIdx.x / SDM_WAVE] = mass;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < SDM_BLOCK / SDM_WAVE; ++w) s += sm[w];
    partial[blockIdx.x] = s;
  }
}

__global__ void k_fold_partials(const double *__restrict__ partial, int64_t n,
                                double *__restrict__ out) {
  // one wave: strided partial sums, then a shuffle tree (order fixed for a given n)
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 64) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[0] = s;
}

extern "C" int sdm_flag_precipitated(sdm_ctx *ctx, const int64_t *cell_origin,
                                     const double *position_in_cell, const double *water_mass,
                                     const int64_t *multiplicity, int64_t *idx, int64_t length,
                                     int64_t n_sd, int n_dims, int64_t *healthy, double level,
                                     const double *displacement, double *rainfall_mass) {
  ARG_TRY(ctx && rainfall_mass && n_dims >= 1 && length >= 0 && length <= n_sd);
  *rainfall_mass = 0.0;
  if (length == 0) return SDM_OK;
  ARG_TRY(cell_origin && position_in_cell && water_mass && multiplicity && idx && healthy &&
          displacement);
  const unsigned nb = grid_for(length);
  int rc = sdm_reserve(ctx, carve_size(sizeof(double) * nb) + 256);
  if (rc) return rc;
  Carver cv(ctx->arena);
  double *partial = cv.take<double>(nb);
  double *out = cv.take<double>(1);
  hipLaunchKernelGGL(k_flag_precipitated, dim3(nb), dim3(SDM_BLOCK), 0, ctx->stream, cell_origin,
                     position_in_cell, water_mass, multiplicity, idx, length, n_sd, n_dims,
                     healthy, level, displacement, partial);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(64), 0, ctx->stream, partial, (int64_t)nb,
                     out);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  memcpy(rainfall_mass, ctx->mailbox, sizeof(double));
  return SDM_OK;
}

// displacement_methods.py:168-190
__global__ void __launch_bounds__(SDM_BLOCK)
k_flag_out_of_column(const int64_t *__restrict__ cell_origin,
                     const double *__restrict__ position_in_cell, int64_t *__restrict__ idx,
                     int64_t length, int64_t n_sd, int n_dims, int64_t *__restrict__ healthy,
                     double top) {
  const int64_t i = TID();
  if (i >= length) return;
  const int64_t last = (int64_t)(n_dims - 1) * n_sd;
  const int64_t k = idx[i];
  const double z = (double)cell_origin[last + k] + position_in_cell[last + k];
  if (z < 0 || z > top) {
    idx[i] = n_sd;
    healthy[0] = 0;
  }
}

extern "C" int sdm_flag_out_of_column(sdm_ctx *ctx, const int64_t *cell_origin,
                                      const double *position_in_cell, int64_t *idx,
                                      int64_t length, int64_t n_sd, int n_dims,
                                      int64_t *healthy, double top) {
  ARG_TRY(ctx && n_dims >= 1 && length >= 0 && length <= n_sd);
  if (length == 0) return SDM_OK;
  ARG_TRY(cell_origin && position_in_cell && idx && healthy);
  hipLaunchKernelGGL(k_flag_out_of_column, GRID1D(length), cell_origin, position_in_cell, idx,
                     length, n_sd, n_dims, healthy, top);
  LAUNCH_CHECK();
  return SDM_OK;
}

// mixed-type Storage ops of Displacement.update_cell_origin (dynamics/displacement.py:146-150):
// an int64 storage takes floor() of a float one, and a float one is decremented by an int64 one
__global__ void __launch_bounds__(SDM_BLOCK)
k_floor_to_i64(int64_t *__restrict__ out, const double *__restrict__ a, int64_t n) {
  const int64_t i = TID();
  if (i < n) out[i] = (int64_t)floor(a[i]);
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_subtract_i64(double *__restrict__ out, const int64_t *__restrict__ b, int64_t n) {
  const int64_t i = TID();
  if (i < n) out[i] -= (double)b[i];
}

extern "C" int sdm_floor_to_i64(sdm_ctx *ctx, int64_t *out, const double *a, int64_t n) {
  ARG_TRY(ctx && n >= 0);
  if (n == 0) return SDM_OK;
  ARG_TRY(out && a);
  hipLaunchKernelGGL(k_floor_to_i64, GRID1D(n), out, a, n);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_subtract_i64(sdm_ctx *ctx, double *out, const int64_t *b, int64_t n) {
  ARG_TRY(ctx && n >= 0);
  if (n == 0) return SDM_OK;
  ARG_TRY(out && b);
  hipLaunchKernelGGL(k_subtract_i64, GRID1D(n), out, b, n);
  LAUNCH_CHECK();
  return SDM_OK;
}
