// Particle displacement (advection + sedimentation) -- the step that precedes the collision
// path in multi-dimensional set-ups and unsorts the state.  Device counterparts of
// PySDM/backends/impl_numba/methods/displacement_methods.py; see include/sdm_hip.h.
#include "common.h"
#include "index.h"

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

// one workgroup: strided partial sums, shuffle tree per wave, the wave sums added in wave order
// (the order is fixed for a given n); accumulate != 0 adds to out[0] instead of overwriting it
__global__ void __launch_bounds__(1024)
k_fold_partials(const double *__restrict__ partial, int64_t n, double *__restrict__ out,
                int accumulate) {
  __shared__ double sm[1024 / SDM_WAVE];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane_id() == 0) sm[threadIdx.x / SDM_WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 1024 / SDM_WAVE; ++w) t += sm[w];
    out[0] = accumulate ? out[0] + t : t;
  }
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
  hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(1024), 0, ctx->stream, partial, (int64_t)nb,
                     out, 0);
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

// ---- the fused step (include/sdm_hip.h: sdm_displacement_step) ---------------------------------
struct DispArgs {
  sdm_disp_cfg cfg;
  sdm_disp_state st;
  double *partial;  // one rainfall partial per workgroup of k_disp_precip
  double *rain;     // [0] running total of the step
  uint8_t *cls;     // per raw super-droplet after the move: 0 stays, 1 precipitates, 2 left the
                    // column -- so that the position-indexed kernels gather one byte, not three
                    // doubles, through idx
};

// A: over raw super-droplets: displacement of every dimension (Arakawa-C interpolation),
// sedimentation, position update -- displacement.py:107-110,123-137
__global__ void __launch_bounds__(SDM_BLOCK) k_disp_move(DispArgs X) {
  const sdm_disp_cfg &c = X.cfg;
  const int64_t k = TID();
  if (k >= c.n_sd) return;
  int64_t origin[3] = {0, 0, 0};
  for (int d = 0; d < c.n_dims; ++d) origin[d] = X.st.cell_origin[d * c.n_sd + k];
  const double n_sub = (double)c.n_substeps;
  for (int dim = 0; dim < c.n_dims; ++dim) {
    int64_t l = 0, r = 0;
    for (int d = 0; d < c.n_dims; ++d) {
      const int64_t extent = c.grid[d] + (d == dim);
      l = l * extent + origin[d];
      r = r * extent + origin[d] + (d == dim);
    }
    const double x = X.st.position_in_cell[dim * c.n_sd + k];
    const double c_l = X.st.courant[dim][l] / n_sub, c_r = X.st.courant[dim][r] / n_sub;
    double v = c_l * (1 - x) + c_r * x;
    if (c.scheme == 0) v = v / (1 - c_r + c_l);
    if (c.enable_sedimentation && dim == c.n_dims - 1) {
      v *= 1 / c.dt_over_dz;
      v -= X.st.fall_velocity[k];
      v *= c.dt_over_dz;
    }
    X.st.displacement[dim * c.n_sd + k] = v;
    const double moved = x + v;
    X.st.position_in_cell[dim * c.n_sd + k] = moved;
    if (dim == c.n_dims - 1) {
      // displacement_methods.py:139-160 and :176-186, evaluated where the operands are at hand;
      // precipitation is tested (and removed) first in the reference, hence takes precedence
      const double z = (double)origin[dim] + moved;
      uint8_t cls = 0;
      if (c.enable_sedimentation && v < 0 && z < c.level) cls = 1;
      else if (z < 0 || z > (double)c.grid[dim]) cls = 2;
      X.cls[k] = cls;
    }
  }
}

// B1: over live positions (grid-stride, DISP_PRECIP_GRID workgroups): precipitation
// (displacement_methods.py:131-166); one partial sum per workgroup, folded by k_fold_partials
#define DISP_PRECIP_GRID 1024
__global__ void __launch_bounds__(SDM_BLOCK) k_disp_precip(DispArgs X) {
  __shared__ double sm[SDM_BLOCK / SDM_WAVE];
  const sdm_disp_cfg &c = X.cfg;
  const int64_t length = X.st.ctl[0];
  double mass = 0.0;
  for (int64_t i = TID(); i < length; i += (int64_t)gridDim.x * SDM_BLOCK) {
    const int64_t k = X.st.idx[i];
    if (X.cls[k] == 1) {
      mass += fabs(X.st.water_mass[k]) * (double)X.st.multiplicity[k];
      X.st.idx[i] = c.n_sd;
      X.st.ctl[3] = 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mass += __shfl_xor(mass, o, 64);
  if (lane_id() == 0) sm[threadIdx.x / SDM_WAVE] = mass;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < SDM_BLOCK / SDM_WAVE; ++w) t += sm[w];
    X.partial[blockIdx.x] = t;
  }
}

// B2: over live positions: out of the column (displacement_methods.py:168-190)
__global__ void __launch_bounds__(SDM_BLOCK) k_disp_column(DispArgs X) {
  const sdm_disp_cfg &c = X.cfg;
  const int64_t i = TID();
  if (i >= X.st.ctl[0]) return;
  const int64_t k = X.st.idx[i];
  if (X.cls[k] == 2) {
    X.st.idx[i] = c.n_sd;
    X.st.ctl[3] = 0;
  }
}

// C: over raw super-droplets: whole cells moved into the cell origin, periodic boundary, cell id
// (displacement.py:143-153, collisions_methods.py:407-416)
__global__ void __launch_bounds__(SDM_BLOCK) k_disp_cells(DispArgs X) {
  const sdm_disp_cfg &c = X.cfg;
  const int64_t k = TID();
  if (k >= c.n_sd) return;
  int64_t id = 0;
  for (int d = 0; d < c.n_dims; ++d) {
    const double x = X.st.position_in_cell[d * c.n_sd + k];
    const int64_t whole = (int64_t)floor(x);
    int64_t o = X.st.cell_origin[d * c.n_sd + k] + whole;
    X.st.position_in_cell[d * c.n_sd + k] = x - (double)whole;
    int64_t m = o % c.grid[d];
    if (m != 0 && ((m < 0) != (c.grid[d] < 0))) m += c.grid[d];  // Python's %
    X.st.cell_origin[d * c.n_sd + k] = m;
    id += m * c.strides[d];
  }
  X.st.cell_id[k] = id;
}

extern "C" int sdm_displacement_step(sdm_ctx *ctx, const sdm_disp_cfg *cfg,
                                     const sdm_disp_state *state, double *rainfall_mass,
                                     int64_t *valid_n_sd) {
  ARG_TRY(ctx && cfg && state && rainfall_mass && valid_n_sd);
  ARG_TRY(cfg->n_sd >= 1 && cfg->n_sd < INT32_MAX && cfg->n_dims >= 1 && cfg->n_dims <= 3);
  ARG_TRY((cfg->scheme == 0 || cfg->scheme == 1) && cfg->n_substeps >= 1);
  for (int d = 0; d < cfg->n_dims; ++d) ARG_TRY(cfg->grid[d] >= 1 && state->courant[d]);
  ARG_TRY(state->displacement && state->position_in_cell && state->cell_origin &&
          state->cell_id && state->water_mass && state->multiplicity && state->idx && state->ctl);
  ARG_TRY(!cfg->enable_sedimentation || (state->fall_velocity && cfg->dt_over_dz != 0));
  const int64_t N = cfg->n_sd;
  const unsigned nb = grid_for(N);
  const size_t need = carve_size(sizeof(double) * nb) + 256 + 256 + carve_size((size_t)N) +
                      sdm_compact_scratch(N);
  int rc = sdm_reserve(ctx, need);
  if (rc) return rc;
  Carver cv(ctx->arena);
  DispArgs X;
  X.cfg = *cfg;
  X.st = *state;
  X.partial = cv.take<double>(nb);
  X.rain = cv.take<double>(1);
  X.cls = cv.take<uint8_t>(N);
  int64_t *cctl = cv.take<int64_t>(8);
  char *compact = cv.take<char>(sdm_compact_scratch(N));
  const unsigned n_precip = nb < DISP_PRECIP_GRID ? nb : DISP_PRECIP_GRID;
  hipStream_t s = ctx->stream;
  const dim3 grid(nb), blk(SDM_BLOCK);
  HIP_TRY(hipMemsetAsync(X.rain, 0, sizeof(double), s));
  for (int sub = 0; sub < cfg->n_substeps; ++sub) {
    hipLaunchKernelGGL(k_disp_move, grid, blk, 0, s, X);
    if (cfg->enable_sedimentation) {
      hipLaunchKernelGGL(k_disp_precip, dim3(n_precip), blk, 0, s, X);
      hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(1024), 0, s, X.partial,
                         (int64_t)n_precip, X.rain, 1);
      rc = sdm_compact_fused_async(ctx, compact, state->multiplicity, state->idx, N, N,
                                   state->ctl, cctl, nullptr, true);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_disp_column, grid, blk, 0, s, X);
    rc = sdm_compact_fused_async(ctx, compact, state->multiplicity, state->idx, N, N, state->ctl,
                                 cctl, nullptr, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_disp_cells, grid, blk, 0, s, X);
    LAUNCH_CHECK();
  }
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, X.rain, sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(ctx->mailbox + 1, state->ctl, sizeof(int64_t) * 8,
                         hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(rainfall_mass, ctx->mailbox, sizeof(double));
  *valid_n_sd = ctx->mailbox[1];
  if (ctx->mailbox[1 + 7] != 0) {
    sdm_set_error("displacement: grid barrier of the compaction kernel timed out");
    return SDM_E_HIP;
  }
  return SDM_OK;
}
