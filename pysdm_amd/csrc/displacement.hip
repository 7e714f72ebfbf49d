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
  // sharded run (sdm_displacement_step_sharded; all NULL otherwise): who is whose (sdm_hip.h:
  // sdm_disp_shard.role), every id's own cell, and where the positions of this process's
  // removed super-droplets are listed
  uint8_t *role;
  int64_t *cell_by_id;
  int64_t *dead;
  double *dead_mass;  // what each listed (precipitated) super-droplet carried
  unsigned long long *n_dead;
  unsigned long long *n_column;  // counted by k_disp_count_column ahead of their listing
};

// A + C: over raw super-droplets: displacement of every dimension (Arakawa-C interpolation),
// sedimentation, position update -- displacement.py:107-110,123-137 -- and, at the end, the cells
__global__ void __launch_bounds__(SDM_BLOCK) k_disp_move(DispArgs X) {
  const sdm_disp_cfg &c = X.cfg;
  const int64_t k = TID();
  if (k >= c.n_sd) return;
  if (X.role && X.role[k] == 0) {  // another process's row (sharded run): not touched
    X.cls[k] = 0;
    return;
  }
  int64_t origin[3] = {0, 0, 0};
  double moved[3] = {0, 0, 0};
  uint8_t cls = 0;
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
    moved[dim] = x + v;
    if (dim == c.n_dims - 1) {
      // displacement_methods.py:139-160 and :176-186, evaluated where the operands are at hand;
      // precipitation is tested (and removed) first in the reference, hence takes precedence
      const double z = (double)origin[dim] + moved[dim];
      if (c.enable_sedimentation && v < 0 && z < c.level) cls = 1;
      else if (z < 0 || z > (double)c.grid[dim]) cls = 2;
      X.cls[k] = cls;
    }
  }
  // C: whole cells moved into the cell origin, periodic boundary, cell id (displacement.py:143-153,
  // collisions_methods.py:407-416).  In the reference this follows the two removals; they read
  // nothing but what `cls` has recorded above, so it is done while the operands are in registers
  // (a pass of its own over the raw columns was 50 us at 2^22)
  int64_t id = 0;
  for (int d = 0; d < c.n_dims; ++d) {
    const double x = moved[d];
    const int64_t whole = (int64_t)floor(x);
    const int64_t o = origin[d] + whole;
    X.st.position_in_cell[d * c.n_sd + k] = x - (double)whole;
    int64_t m = o % c.grid[d];
    if (m != 0 && ((m < 0) != (c.grid[d] < 0))) m += c.grid[d];  // Python's %
    X.st.cell_origin[d * c.n_sd + k] = m;
    id += m * c.strides[d];
  }
  if (X.role) {
    // (a removed one, role 2 or about to be, may stand in the permutation as a placeholder for
    // somebody else's super-droplet: its cell_id entry then belongs to that position)
    X.cell_by_id[k] = id;
    if (X.role[k] != 1 || cls != 0) return;
  }
  X.st.cell_id[k] = id;
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
    if (X.cls[k] == 1 && (!X.role || X.role[k] == 1)) {
      const double carried = fabs(X.st.water_mass[k]) * (double)X.st.multiplicity[k];
      if (X.role) {
        // sharded: position and mass are announced first; every process flags the position and
        // adds the masses up in the one-process order (k_disp_rain below)
        X.role[k] = 2;
        const unsigned long long at = atomicAdd(X.n_dead, 1ull);
        X.dead[at] = i;
        X.dead_mass[at] = carried;
      } else {
        mass += carried;
        X.st.idx[i] = c.n_sd;
        X.st.ctl[3] = 0;
      }
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

// sharded: how many of this process's super-droplets k_disp_column is going to list, known as soon
// as k_disp_move has classified them - so that ONE exchange of counts serves both removals
__global__ void __launch_bounds__(SDM_BLOCK) k_disp_count_column(DispArgs X) {
  const int64_t k = TID();
  const bool leaves = k < X.cfg.n_sd && X.role[k] == 1 && X.cls[k] == 2;
  const unsigned long long m = __ballot(leaves);
  if (m != 0 && lane_id() == __ffsll((long long)m) - 1)
    atomicAdd(X.n_column, (unsigned long long)__popcll(m));
}

// B2: over live positions: out of the column (displacement_methods.py:168-190)
__global__ void __launch_bounds__(SDM_BLOCK) k_disp_column(DispArgs X) {
  const sdm_disp_cfg &c = X.cfg;
  const int64_t i = TID();
  if (i >= X.st.ctl[0]) return;
  const int64_t k = X.st.idx[i];
  if (X.cls[k] == 2 && (!X.role || X.role[k] == 1)) {
    if (X.role) {
      X.role[k] = 2;
      X.dead[atomicAdd(X.n_dead, 1ull)] = i;
    } else {
      X.st.idx[i] = c.n_sd;
      X.st.ctl[3] = 0;
    }
  }
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
  X.role = nullptr;
  X.cell_by_id = nullptr;
  X.dead = nullptr;
  X.dead_mass = nullptr;
  X.n_dead = nullptr;
  X.n_column = nullptr;
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

// ---- the displacement step of a sharded run (include/sdm_hip.h: sdm_disp_shard) ----------------
// The kernels above with `role` set, and what crosses the processes in between.  Every list is
// filled through an atomic counter: its order differs from run to run and does not matter (the
// positions of the removed are a set; the placeholders that trade places are interchangeable).
struct ShardLists {
  int64_t *words;      // the exchange buffer: [2 x all movers][row x all that changed owner]
  int64_t row;         // words per row
  int64_t tot_a, tot_b;
  int n_dims, n_attr;
  int64_t n_sd;
};

__global__ void __launch_bounds__(SDM_BLOCK)
k_role_init(uint8_t *__restrict__ role, const uint8_t *__restrict__ owned,
            const int64_t *__restrict__ cell_by_id, int64_t n_sd) {
  const int64_t k = TID();
  if (k < n_sd) role[k] = owned[cell_by_id[k]] ? 2 : 0;  // (2: not in the permutation = removed)
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_role_alive(uint8_t *__restrict__ role, const int64_t *__restrict__ idx,
             const int64_t *__restrict__ ctl) {
  const int64_t i = TID();
  if (i < ctl[0] && role[idx[i]]) role[idx[i]] = 1;
}
// what a collision step removed since the last call (the owner sees the zero multiplicity; its
// compaction took the id out of the permutation), and the cells the call begins with
__global__ void __launch_bounds__(SDM_BLOCK)
k_shard_begin(uint8_t *__restrict__ role, const int64_t *__restrict__ multiplicity,
              const int64_t *__restrict__ cell_by_id, int64_t *__restrict__ cell0, int64_t n_sd,
              double *__restrict__ rain, unsigned long long *__restrict__ counters) {
  const int64_t k = TID();
  if (k == 0) *rain = 0.0;       // (the call's rainfall sum and its eight list counters start
  if (k < 8) counters[k] = 0;    //  from zero: no fills of their own)
  if (k >= n_sd) return;
  if (role[k] == 1 && multiplicity[k] == 0) role[k] = 2;
  cell0[k] = cell_by_id[k];
}
// counts[l * world + r] = how many this process is about to put on list l (its own slot r only);
// mine[l]: the same for the host
// (the counters of the two removal lists start from zero again afterwards: the lists' lengths are
// with the host from here on)
struct FourCounts { unsigned long long *n[4]; };
__global__ void k_pack_counts(double *__restrict__ counts, int world, int rank, int lists,
                              FourCounts C, unsigned long long *__restrict__ mine) {
  const int t = threadIdx.x, l = t / world, r = t % world;
  __shared__ unsigned long long v[4];
  if (t < 4) v[t] = t < lists ? *C.n[t] : 0;
  __syncthreads();
  if (l < lists) counts[t] = r == rank ? (double)v[l] : 0.0;
  if (t < lists) mine[t] = v[t];
  if (t < 2) *C.n[t] = 0;
}
// this process's slices of an exchange of lists, the rest of the `words` zero (a sum over the
// processes then IS the concatenation): own positions at [before, before + mine), with rain their
// masses (bit patterns) at total + the same
__global__ void __launch_bounds__(SDM_BLOCK)
k_place_slices(int64_t *__restrict__ out, int64_t words, int64_t total, int64_t before,
               int64_t mine, const int64_t *__restrict__ dead,
               const int64_t *__restrict__ dead_mass) {
  const int64_t i = TID();
  if (i >= words) return;
  const int64_t j = i < total ? i : i - total;
  int64_t v = 0;
  if (j >= before && j < before + mine) v = (i < total ? dead : dead_mass)[j - before];
  out[i] = v;
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_rain_unscatter(double *__restrict__ carried, const int64_t *__restrict__ words, int64_t total) {
  const int64_t j = TID();
  if (j < total) carried[words[j]] = 0.0;
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_copy_i64(int64_t *__restrict__ out, const int64_t *__restrict__ in, int64_t n) {
  const int64_t i = TID();
  if (i < n) out[i] = in[i];
}
// (n_dead: the counter of the list just consumed starts from zero again)
__global__ void __launch_bounds__(SDM_BLOCK)
k_flag_positions(int64_t *__restrict__ idx, const int64_t *__restrict__ dead, int64_t n,
                 int64_t n_sd, int64_t *__restrict__ ctl, unsigned long long *__restrict__ n_dead) {
  const int64_t i = TID();
  if (i >= n) return;
  idx[dead[i]] = n_sd;
  if (i == 0) { ctl[3] = 0; *n_dead = 0; }
}
// the rainfall of a sharded sub-step, to the bits of the one-process run: the masses of the
// precipitated (from all processes) scattered to their positions in an otherwise zero array, then
// summed exactly as k_disp_precip sums them - same grid, same strides, same trees; x + 0.0 = x
__global__ void __launch_bounds__(SDM_BLOCK)
k_rain_scatter(double *__restrict__ carried, const int64_t *__restrict__ words, int64_t total) {
  const int64_t j = TID();
  if (j < total) carried[words[j]] = __longlong_as_double(words[total + j]);
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_disp_rain(const double *__restrict__ carried, const int64_t *__restrict__ ctl,
            double *__restrict__ partial) {
  __shared__ double sm[SDM_BLOCK / SDM_WAVE];
  const int64_t length = ctl[0];
  double mass = 0.0;
  for (int64_t i = TID(); i < length; i += (int64_t)gridDim.x * SDM_BLOCK) mass += carried[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mass += __shfl_xor(mass, o, 64);
  if (lane_id() == 0) sm[threadIdx.x / SDM_WAVE] = mass;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < SDM_BLOCK / SDM_WAVE; ++w) t += sm[w];
    partial[blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_inverse(int32_t *__restrict__ inv, const int64_t *__restrict__ idx,
          const int64_t *__restrict__ ctl) {
  const int64_t i = TID();
  if (i < ctl[0]) inv[idx[i]] = (int32_t)i;
}
// movers: this process's rows (alive or removed) whose cell changed (a); of the alive ones, those
// whose new cell is another process's (b).  Listed by stream compaction - per-workgroup counts,
// one scan, ranks inside the workgroup from ballots - so that the lists come out in id order and
// nobody queues at one counter (65 k same-address atomics were 0.8 ms at 2^22)
// `cls`: counting ahead of the removals of the last sub-step (whoever k_disp_move classified as
// leaving will be a removed one by the time the lists are written); NULL: as things are
__device__ __forceinline__ void mover_kind(const uint8_t *__restrict__ role,
                                           const uint8_t *__restrict__ cls,
                                           const uint8_t *__restrict__ owned,
                                           const int64_t *__restrict__ cell_by_id,
                                           const int64_t *__restrict__ cell0, int64_t n_sd,
                                           int64_t k, bool *a, bool *b) {
  *a = *b = false;
  if (k >= n_sd || role[k] == 0 || cell_by_id[k] == cell0[k]) return;
  const bool alive = role[k] == 1 && (!cls || cls[k] == 0);
  // (a removed one's cell is read by `normalize` alone, as cell_id[pair number]: ids from
  // (n_sd + 1) / 2 on are never asked for, and a quarter of a long run's traffic was theirs)
  if (alive || k < (n_sd + 1) / 2) {
    *a = true;
    *b = alive && !owned[cell_by_id[k]];
  }
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_count_movers(const uint8_t *__restrict__ role, const uint8_t *__restrict__ cls,
               const uint8_t *__restrict__ owned, const int64_t *__restrict__ cell_by_id,
               const int64_t *__restrict__ cell0, int64_t n_sd, int32_t *__restrict__ blk_a,
               int32_t *__restrict__ blk_b) {
  __shared__ int sa[SDM_BLOCK / SDM_WAVE], sb[SDM_BLOCK / SDM_WAVE];
  bool a, b;
  mover_kind(role, cls, owned, cell_by_id, cell0, n_sd, TID(), &a, &b);
  const unsigned long long ma = __ballot(a), mb = __ballot(b);
  if (lane_id() == 0) {
    sa[threadIdx.x / SDM_WAVE] = __popcll(ma);
    sb[threadIdx.x / SDM_WAVE] = __popcll(mb);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int ta = 0, tb = 0;
    for (int w = 0; w < SDM_BLOCK / SDM_WAVE; ++w) { ta += sa[w]; tb += sb[w]; }
    blk_a[blockIdx.x] = ta;
    blk_b[blockIdx.x] = tb;
  }
}
// one workgroup: exclusive prefix sums of the per-workgroup counts, in place; totals -> n[0], n[1]
// (the counts go through LDS - read and written back coalesced -, a thread sums a run of
// consecutive ones there, one scan over the 1024 run totals: the first version scanned 1024
// counts per round with three barriers each, 38 us for 16 k counts)
#define SCAN_MOVERS_CAP 16384  // counts held in LDS (2^22 super-droplets); more: in rounds
#define SCAN_MOVERS_PER (SCAN_MOVERS_CAP / 1024)
// (a thread's run of SCAN_MOVERS_PER counts stands at a stride of PER + 1 words: no bank conflicts)
#define SCAN_MOVERS_AT(i) ((i) + (i) / SCAN_MOVERS_PER)
#define SCAN_MOVERS_LDS (2 * (SCAN_MOVERS_CAP + 1024) * sizeof(int32_t))
__global__ void __launch_bounds__(1024)
k_scan_movers(int32_t *__restrict__ blk_a, int32_t *__restrict__ blk_b, int64_t nb,
              unsigned long long *__restrict__ n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t *la = (int32_t *)smem, *lb = la + SCAN_MOVERS_CAP + 1024;
  __shared__ int sa[1024], sb[1024];
  long long carry_a = 0, carry_b = 0;
  for (int64_t base = 0; base < nb; base += SCAN_MOVERS_CAP) {
    const int m = (int)(nb - base < SCAN_MOVERS_CAP ? nb - base : SCAN_MOVERS_CAP);
    for (int i = threadIdx.x; i < SCAN_MOVERS_CAP; i += 1024) {
      la[SCAN_MOVERS_AT(i)] = i < m ? blk_a[base + i] : 0;
      lb[SCAN_MOVERS_AT(i)] = i < m ? blk_b[base + i] : 0;
    }
    __syncthreads();
    const int i0 = threadIdx.x * SCAN_MOVERS_PER;
    int ta = 0, tb = 0;
#pragma unroll
    for (int k = 0; k < SCAN_MOVERS_PER; ++k) {
      ta += la[SCAN_MOVERS_AT(i0 + k)];
      tb += lb[SCAN_MOVERS_AT(i0 + k)];
    }
    // inclusive scan of the 1024 run totals: within the wavefront by shuffles, then over the 16
    int ia = ta, ib = tb;
#pragma unroll
    for (int o = 1; o < SDM_WAVE; o <<= 1) {
      const int xa = __shfl_up(ia, o, 64), xb = __shfl_up(ib, o, 64);
      if (lane_id() >= o) { ia += xa; ib += xb; }
    }
    if (lane_id() == SDM_WAVE - 1) { sa[threadIdx.x / SDM_WAVE] = ia; sb[threadIdx.x / SDM_WAVE] = ib; }
    __syncthreads();
    int wa = 0, wb = 0, all_a = 0, all_b = 0;
    for (int w = 0; w < 1024 / SDM_WAVE; ++w) {
      if (w < (int)(threadIdx.x / SDM_WAVE)) { wa += sa[w]; wb += sb[w]; }
      all_a += sa[w];
      all_b += sb[w];
    }
    long long ra = carry_a + wa + ia - ta, rb = carry_b + wb + ib - tb;
#pragma unroll
    for (int k = 0; k < SCAN_MOVERS_PER; ++k) {
      const int va = la[SCAN_MOVERS_AT(i0 + k)], vb = lb[SCAN_MOVERS_AT(i0 + k)];
      la[SCAN_MOVERS_AT(i0 + k)] = (int32_t)ra;
      lb[SCAN_MOVERS_AT(i0 + k)] = (int32_t)rb;
      ra += va;
      rb += vb;
    }
    carry_a += all_a;
    carry_b += all_b;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 1024) {
      blk_a[base + i] = la[SCAN_MOVERS_AT(i)];
      blk_b[base + i] = lb[SCAN_MOVERS_AT(i)];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    n[0] = (unsigned long long)carry_a;
    n[1] = (unsigned long long)carry_b;
  }
}
// at_a, at_b: where this process's slices begin (in entries); blk_*: the scanned counts
__global__ void __launch_bounds__(SDM_BLOCK)
k_list_movers(ShardLists L, uint8_t *__restrict__ role, const uint8_t *__restrict__ owned,
              const int64_t *__restrict__ cell_by_id, const int64_t *__restrict__ cell0,
              const int32_t *__restrict__ inv, const int64_t *__restrict__ multiplicity,
              const double *__restrict__ attributes, const int64_t *__restrict__ cell_origin,
              const double *__restrict__ position_in_cell, const int32_t *__restrict__ blk_a,
              const int32_t *__restrict__ blk_b, int64_t at_a, int64_t at_b) {
  __shared__ int sa[SDM_BLOCK / SDM_WAVE], sb[SDM_BLOCK / SDM_WAVE];
  const int64_t k = TID();
  bool a, b;
  mover_kind(role, nullptr, owned, cell_by_id, cell0, L.n_sd, k, &a, &b);
  const unsigned long long ma = __ballot(a), mb = __ballot(b);
  const unsigned long long below = (1ull << lane_id()) - 1;
  if (lane_id() == 0) {
    sa[threadIdx.x / SDM_WAVE] = __popcll(ma);
    sb[threadIdx.x / SDM_WAVE] = __popcll(mb);
  }
  __syncthreads();
  if (!a) return;
  int64_t ra = at_a + blk_a[blockIdx.x] + __popcll(ma & below);
  int64_t rb = at_b + blk_b[blockIdx.x] + __popcll(mb & below);
  for (int w = 0; w < (int)(threadIdx.x / SDM_WAVE); ++w) { ra += sa[w]; rb += sb[w]; }
  const int64_t to = cell_by_id[k];
  const int64_t p = role[k] == 1 ? (int64_t)inv[k] : -1;
  // a changed cell travels as two words: (position + 1) << 32 | id, new cell (n_sd < 2^31)
  int64_t *e = L.words + 2 * ra;
  e[0] = ((p + 1) << 32) | k;
  e[1] = to;
  if (!b) return;
  role[k] = 0;  // it goes on as a placeholder here
  int64_t *w = L.words + 2 * L.tot_a + L.row * rb;
  w[0] = p;
  w[1] = k;
  w[2] = to;
  w[3] = multiplicity[k];
  for (int d = 0; d < L.n_dims; ++d) w[4 + d] = cell_origin[d * L.n_sd + k];
  for (int x = 0; x < L.n_attr; ++x)
    w[4 + L.n_dims + x] = __double_as_longlong(attributes[x * L.n_sd + k]);
  for (int d = 0; d < L.n_dims; ++d)
    w[4 + L.n_dims + L.n_attr + d] = __double_as_longlong(position_in_cell[d * L.n_sd + k]);
}
// slot of an entry appended to a list through its counter: one atomic per wavefront, not per lane
// (every lane of the wavefront must call; `take` = this lane appends)
__device__ __forceinline__ unsigned long long wave_append(unsigned long long *counter, bool take) {
  const unsigned long long mask = __ballot(take);
  if (mask == 0) return 0;
  const int leader = __ffsll((long long)mask) - 1;
  unsigned long long base = 0;
  if (lane_id() == leader) base = atomicAdd(counter, (unsigned long long)__popcll(mask));
  base = __shfl((long long)base, leader, 64);
  return base + __popcll(mask & ((1ull << lane_id()) - 1));
}

// arrivals: rows whose new cell is this process's.  The true id goes to the true position; the
// placeholders involved trade places, each taking the cell id of the position it moves to
__global__ void __launch_bounds__(SDM_BLOCK)
k_arrivals_mark(ShardLists L, const uint8_t *__restrict__ owned, uint8_t *__restrict__ is_p,
                uint8_t *__restrict__ is_x, unsigned long long *__restrict__ n_arrived) {
  const int64_t j = TID();
  const int64_t *w = L.words + 2 * L.tot_a + L.row * (j < L.tot_b ? j : 0);
  const bool mine = j < L.tot_b && owned[w[2]];
  if (mine) {
    is_p[w[0]] = 1;
    is_x[w[1]] = 1;
  }
  const unsigned long long m = __ballot(mine);
  if (m != 0 && lane_id() == __ffsll((long long)m) - 1)
    atomicAdd(n_arrived, (unsigned long long)__popcll(m));
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_arrivals_lists(ShardLists L, const uint8_t *__restrict__ owned,
                 const uint8_t *__restrict__ is_p, const uint8_t *__restrict__ is_x,
                 const int32_t *__restrict__ inv, const int64_t *__restrict__ idx,
                 const int64_t *__restrict__ cell_id, int32_t *__restrict__ free_slot,
                 int64_t *__restrict__ free_cell, int32_t *__restrict__ homeless,
                 unsigned long long *__restrict__ n) {  // n[0] free slots, n[1] homeless ids
  const int64_t j = TID();
  const int64_t *w = L.words + 2 * L.tot_a + L.row * (j < L.tot_b ? j : 0);
  const bool mine = j < L.tot_b && owned[w[2]];
  const int32_t at = mine ? inv[w[1]] : -1;
  const bool frees = mine && at >= 0 && !is_p[at];
  const unsigned long long f = wave_append(&n[0], frees);
  if (frees) {
    free_slot[f] = at;
    free_cell[f] = cell_id[w[1]];
  }
  const int64_t there = mine ? idx[w[0]] : 0;
  const bool loses = mine && !is_x[there];
  const unsigned long long h = wave_append(&n[1], loses);
  if (loses) homeless[h] = (int32_t)there;
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_arrivals_place(ShardLists L, const uint8_t *__restrict__ owned, int64_t *__restrict__ idx) {
  const int64_t j = TID();
  if (j >= L.tot_b) return;
  const int64_t *w = L.words + 2 * L.tot_a + L.row * j;
  if (owned[w[2]]) idx[w[0]] = w[1];
}
// (the homeless beyond the free slots leave the live set of this process: their places went to
// arrivals that were not among its placeholders)
__global__ void __launch_bounds__(SDM_BLOCK)
k_arrivals_fill(int64_t bound, const unsigned long long *__restrict__ n,
                const int32_t *__restrict__ free_slot, const int64_t *__restrict__ free_cell,
                const int32_t *__restrict__ homeless, int64_t *__restrict__ idx,
                int64_t *__restrict__ cell_id) {
  const int64_t j = TID();
  if (j >= bound || j >= (int64_t)n[0]) return;
  idx[free_slot[j]] = homeless[j];
  cell_id[homeless[j]] = free_cell[j];
}
// everybody's list of changed cells: the id's own cell, and the cell id of whatever id stands at
// that position here
__global__ void __launch_bounds__(SDM_BLOCK)
k_apply_cells(ShardLists L, const int64_t *__restrict__ idx, int64_t *__restrict__ cell_id,
              int64_t *__restrict__ cell_by_id) {
  const int64_t j = TID();
  if (j >= L.tot_a) return;
  const int64_t *a = L.words + 2 * j;
  const int64_t at = (a[0] >> 32) - 1, id = a[0] & 0xffffffffLL;
  cell_by_id[id] = a[1];
  if (at >= 0) cell_id[idx[at]] = a[1];
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_apply_rows(ShardLists L, const uint8_t *__restrict__ owned, uint8_t *__restrict__ role,
             int64_t *__restrict__ multiplicity, double *__restrict__ attributes,
             int64_t *__restrict__ cell_origin, double *__restrict__ position_in_cell) {
  const int64_t j = TID();
  if (j >= L.tot_b) return;
  const int64_t *w = L.words + 2 * L.tot_a + L.row * j;
  if (!owned[w[2]]) return;
  const int64_t k = w[1];
  role[k] = 1;
  multiplicity[k] = w[3];
  for (int d = 0; d < L.n_dims; ++d) cell_origin[d * L.n_sd + k] = w[4 + d];
  for (int x = 0; x < L.n_attr; ++x)
    attributes[x * L.n_sd + k] = __longlong_as_double(w[4 + L.n_dims + x]);
  for (int d = 0; d < L.n_dims; ++d)
    position_in_cell[d * L.n_sd + k] = __longlong_as_double(w[4 + L.n_dims + L.n_attr + d]);
}

__global__ void k_disp_final(const int64_t *__restrict__ arrived, const int64_t *__restrict__ ctl,
                             const double *__restrict__ rain, int64_t *__restrict__ out) {
  const int t = threadIdx.x;
  if (t == 0) out[0] = arrived[0];
  if (t >= 1 && t < 9) out[t] = ctl[t - 1];
  if (t == 9) out[9] = __double_as_longlong(rain[0]);
}

extern "C" int sdm_displacement_step_sharded(sdm_ctx *ctx, const sdm_disp_cfg *cfg,
                                             const sdm_disp_state *state, sdm_disp_shard *sh,
                                             double *rainfall_mass, int64_t *valid_n_sd) {
  ARG_TRY(ctx && cfg && state && sh && rainfall_mass && valid_n_sd);
  ARG_TRY(cfg->n_sd >= 1 && cfg->n_sd < INT32_MAX && cfg->n_dims >= 1 && cfg->n_dims <= 3);
  ARG_TRY((cfg->scheme == 0 || cfg->scheme == 1) && cfg->n_substeps >= 1);
  for (int d = 0; d < cfg->n_dims; ++d) ARG_TRY(cfg->grid[d] >= 1 && state->courant[d]);
  ARG_TRY(state->displacement && state->position_in_cell && state->cell_origin &&
          state->cell_id && state->water_mass && state->multiplicity && state->idx && state->ctl);
  ARG_TRY(!cfg->enable_sedimentation || (state->fall_velocity && cfg->dt_over_dz != 0));
  ARG_TRY(sh->cell_owned && (sh->exchange || ctx->comm) && sh->xchg_counts && sh->xchg_words &&
          sh->cell_id_by_id && sh->role && sh->multiplicity && sh->attributes);
  ARG_TRY(sh->n_attr >= 1 && sh->shard_world >= 1 && sh->shard_world <= 256 &&
          sh->shard_rank >= 0 && sh->shard_rank < sh->shard_world && sh->word_capacity >= 0);
  const int64_t N = cfg->n_sd;
  const int D = cfg->n_dims, W = sh->shard_world, R = sh->shard_rank;
  const unsigned nb = grid_for(N);
  const size_t need = carve_size(sizeof(double) * nb) + 256 + 256 + carve_size((size_t)N) +
                      sdm_compact_scratch(N) + 2 * carve_size(sizeof(int64_t) * (size_t)N) +
                      carve_size(sizeof(int32_t) * (size_t)N) + carve_size(2 * (size_t)N) +
                      2 * carve_size(sizeof(int32_t) * (size_t)N) +
                      2 * carve_size(sizeof(int64_t) * (size_t)N) +
                      2 * carve_size(sizeof(int32_t) * (size_t)nb) + 2048;
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
  int64_t *cell0 = cv.take<int64_t>(N);
  X.dead = cv.take<int64_t>(N);
  X.dead_mass = cv.take<double>(N);
  int32_t *inv = cv.take<int32_t>(N);
  uint8_t *mark = cv.take<uint8_t>(2 * (size_t)N);
  int32_t *free_slot = cv.take<int32_t>(N), *homeless = cv.take<int32_t>(N);
  int64_t *free_cell = cv.take<int64_t>(N);
  int32_t *blk_a = cv.take<int32_t>(nb), *blk_b = cv.take<int32_t>(nb);
  unsigned long long *counters = cv.take<unsigned long long>(8);
  X.n_dead = counters;  // [0]; [2], [3]: movers; [6]: arrivals; then [0], [1] again (free / homeless)
  X.n_column = counters + 7;
  X.role = sh->role;
  X.cell_by_id = sh->cell_id_by_id;
  const unsigned n_precip = nb < DISP_PRECIP_GRID ? nb : DISP_PRECIP_GRID;
  hipStream_t s = ctx->stream;
  const dim3 grid(nb), blk(SDM_BLOCK), one(1);
  sh->n_moved = sh->n_left = sh->n_arrived = sh->n_words = sh->n_removed = 0;
  if (!sh->role_ready) {
    hipLaunchKernelGGL(k_role_init, grid, blk, 0, s, sh->role, sh->cell_owned, sh->cell_id_by_id, N);
    hipLaunchKernelGGL(k_role_alive, grid, blk, 0, s, sh->role, state->idx, state->ctl);
    LAUNCH_CHECK();
    sh->role_ready = 1;
  }
  hipLaunchKernelGGL(k_shard_begin, grid, blk, 0, s, sh->role, (const int64_t *)sh->multiplicity,
                     (const int64_t *)sh->cell_id_by_id, cell0, N, X.rain, counters);
  LAUNCH_CHECK();
  double host_counts[4 * 256];
  // the precipitated masses by position, for the rainfall sum (k_disp_rain): an array of the
  // context's own, all zero between uses (scratch of the arena is anybody's between calls)
  if (ctx->rain_carry_len < N) {
    if (ctx->rain_carry) (void)hipFree(ctx->rain_carry);
    ctx->rain_carry = nullptr;
    ctx->rain_carry_len = 0;
    HIP_TRY(hipMalloc((void **)&ctx->rain_carry, sizeof(double) * (size_t)N));
    HIP_TRY(hipMemsetAsync(ctx->rain_carry, 0, sizeof(double) * (size_t)N, ctx->stream));
    ctx->rain_carry_len = N;
  }
  double *carried = ctx->rain_carry;
  // ONE exchange of counts per sub-step: both removals (who leaves the column is decided by the
  // same classification as who precipitates) and, in the last sub-step, the movers too (their
  // cells are known as soon as k_disp_move has run).  Totals and this process's offsets, per list
  int64_t total4[4] = {0, 0, 0, 0}, before4[4] = {0, 0, 0, 0}, mine4[4] = {0, 0, 0, 0};
  unsigned long long *mine_dev = cv.take<unsigned long long>(4);
  int64_t *final10 = cv.take<int64_t>(10);
  auto exchange_counts = [&](int lists) -> int {
    FourCounts C4 = {{X.n_dead, X.n_column, counters + 2, counters + 3}};
    hipLaunchKernelGGL(k_pack_counts, one, dim3(1024), 0, s, sh->xchg_counts, W, R, lists, C4,
                       mine_dev);
    LAUNCH_CHECK();
    if (sdm_exchange(ctx, sh->exchange, sh->exchange_user, SDM_XCHG_SUM_F64, sh->xchg_counts,
                     lists * W) != 0) {
      return SDM_E_HIP;
    }
    HIP_TRY(hipMemcpyAsync(host_counts, sh->xchg_counts, sizeof(double) * (size_t)(lists * W),
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int l = 0; l < lists; ++l) {
      mine4[l] = (int64_t)host_counts[l * W + R];  // (slot R of the sum: this process's alone)
      total4[l] = before4[l] = 0;
      for (int r = 0; r < W; ++r) {
        if (r < R) before4[l] += (int64_t)host_counts[l * W + r];
        total4[l] += (int64_t)host_counts[l * W + r];
      }
    }
    return SDM_OK;
  };
  // the positions listed by k_disp_precip / k_disp_column on every process -> flagged on every
  // process -> the reference's compaction on every process's own permutation
  auto remove_listed = [&](bool with_rain, int64_t total, int64_t before, int64_t mine) -> int {
    // (nothing to remove anywhere: nothing to launch - the one-process step adds this sub-step's
    // rainfall of 0.0 to its sum, which changes no bit of it)
    if (total == 0) return SDM_OK;
    const int64_t words = with_rain ? 2 * total : total;
    if (words > sh->word_capacity || total > N || before + mine > total) {
      sdm_set_error("sharded displacement: word_capacity too small for %lld removed",
                    (long long)total);
      return SDM_E_ARG;
    }
    hipLaunchKernelGGL(k_place_slices, dim3(grid_for(words)), blk, 0, s, sh->xchg_words, words,
                       total, before, mine, (const int64_t *)X.dead, (const int64_t *)X.dead_mass);
    LAUNCH_CHECK();
    if (sdm_exchange(ctx, sh->exchange, sh->exchange_user, SDM_XCHG_SUM_I64, sh->xchg_words,
                     words) != 0) {
      return SDM_E_HIP;
    }
    sh->n_words += words;
    sh->n_removed += total;
    if (with_rain) {
      // (`carried` is zero between uses: the entries scattered are taken back afterwards)
      hipLaunchKernelGGL(k_rain_scatter, dim3(grid_for(total)), blk, 0, s, carried,
                         (const int64_t *)sh->xchg_words, total);
      hipLaunchKernelGGL(k_disp_rain, dim3(n_precip), blk, 0, s, (const double *)carried,
                         (const int64_t *)state->ctl, X.partial);
      hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(1024), 0, s, X.partial,
                         (int64_t)n_precip, X.rain, 1);
      hipLaunchKernelGGL(k_rain_unscatter, dim3(grid_for(total)), blk, 0, s, carried,
                         (const int64_t *)sh->xchg_words, total);
      LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_flag_positions, dim3(grid_for(total)), blk, 0, s, state->idx,
                       (const int64_t *)sh->xchg_words, total, N, state->ctl, X.n_dead);
    LAUNCH_CHECK();
    // the positions to remove are a list: no pass over the permutation (index.hip)
    if (sdm_compact_listed_fits(total))
      return sdm_compact_listed_async(ctx, compact, state->idx, sh->xchg_words, total, N, N,
                                      state->ctl, cctl, nullptr);
    return sdm_compact_fused_async(ctx, compact, state->multiplicity, state->idx, N, N,
                                   state->ctl, cctl, nullptr, true);
  };
  for (int sub = 0; sub < cfg->n_substeps; ++sub) {
    const bool last_sub = sub == cfg->n_substeps - 1;
    hipLaunchKernelGGL(k_disp_move, grid, blk, 0, s, X);
    hipLaunchKernelGGL(k_disp_count_column, grid, blk, 0, s, X);
    if (cfg->enable_sedimentation) hipLaunchKernelGGL(k_disp_precip, dim3(n_precip), blk, 0, s, X);
    if (last_sub) {  // the movers of the call, counted ahead of this sub-step's removals
      hipLaunchKernelGGL(k_count_movers, grid, blk, 0, s, (const uint8_t *)sh->role,
                         (const uint8_t *)X.cls, sh->cell_owned,
                         (const int64_t *)sh->cell_id_by_id, (const int64_t *)cell0, N, blk_a,
                         blk_b);
      // (> 64 KB of dynamic LDS must be opted into)
      HIP_TRY(hipFuncSetAttribute((const void *)k_scan_movers,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)SCAN_MOVERS_LDS));
      hipLaunchKernelGGL(k_scan_movers, one, dim3(1024), SCAN_MOVERS_LDS, s,
                         blk_a, blk_b, (int64_t)nb, counters + 2);
    }
    LAUNCH_CHECK();
    rc = exchange_counts(last_sub ? 4 : 2);
    if (rc) return rc;
    if (cfg->enable_sedimentation) {
      rc = remove_listed(true, total4[0], before4[0], mine4[0]);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_disp_column, grid, blk, 0, s, X);
    LAUNCH_CHECK();
    rc = remove_listed(false, total4[1], before4[1], mine4[1]);
    if (rc) return rc;
  }
  // ---- who changed cell, who changed owner ----------------------------------------------------
  ShardLists L;
  L.words = sh->xchg_words;
  L.row = 4 + D + sh->n_attr + D;
  L.n_dims = D;
  L.n_attr = sh->n_attr;
  L.n_sd = N;
  HIP_TRY(hipMemsetAsync(inv, 0xff, sizeof(int32_t) * (size_t)N, s));
  hipLaunchKernelGGL(k_inverse, grid, blk, 0, s, inv, (const int64_t *)state->idx,
                     (const int64_t *)state->ctl);
  sh->n_moved = mine4[2];
  sh->n_left = mine4[3];
  const int64_t tot_a = total4[2], tot_b = total4[3], at_a = before4[2], at_b = before4[3];
  L.tot_a = tot_a;
  L.tot_b = tot_b;
  const int64_t words = 2 * tot_a + L.row * tot_b;
  if (words > sh->word_capacity || tot_a > 2 * N || tot_b > N) {
    sdm_set_error("sharded displacement: word_capacity too small (%lld words needed)",
                  (long long)words);
    return SDM_E_ARG;
  }
  if (words > 0) {
    HIP_TRY(hipMemsetAsync(sh->xchg_words, 0, sizeof(int64_t) * (size_t)words, s));
    hipLaunchKernelGGL(k_list_movers, grid, blk, 0, s, L, sh->role, sh->cell_owned,
                       (const int64_t *)sh->cell_id_by_id, (const int64_t *)cell0,
                       (const int32_t *)inv, (const int64_t *)sh->multiplicity,
                       (const double *)sh->attributes, (const int64_t *)state->cell_origin,
                       (const double *)state->position_in_cell, (const int32_t *)blk_a,
                       (const int32_t *)blk_b, at_a, at_b);
    LAUNCH_CHECK();
    if (sdm_exchange(ctx, sh->exchange, sh->exchange_user, SDM_XCHG_SUM_I64, sh->xchg_words,
                     words) != 0) {
      return SDM_E_HIP;
    }
    sh->n_words += words;
    if (tot_b > 0) {
      const dim3 gb(grid_for(tot_b));
      HIP_TRY(hipMemsetAsync(mark, 0, 2 * (size_t)N, s));
      HIP_TRY(hipMemsetAsync(counters, 0, sizeof(unsigned long long) * 2, s));
      HIP_TRY(hipMemsetAsync(counters + 6, 0, sizeof(unsigned long long), s));
      hipLaunchKernelGGL(k_arrivals_mark, gb, blk, 0, s, L, sh->cell_owned, mark, mark + N,
                         counters + 6);
      hipLaunchKernelGGL(k_arrivals_lists, gb, blk, 0, s, L, sh->cell_owned,
                         (const uint8_t *)mark, (const uint8_t *)(mark + N), (const int32_t *)inv,
                         (const int64_t *)state->idx, (const int64_t *)state->cell_id, free_slot,
                         free_cell, homeless, counters);
      hipLaunchKernelGGL(k_arrivals_place, gb, blk, 0, s, L, sh->cell_owned, state->idx);
      hipLaunchKernelGGL(k_arrivals_fill, gb, blk, 0, s, tot_b,
                         (const unsigned long long *)counters, (const int32_t *)free_slot,
                         (const int64_t *)free_cell, (const int32_t *)homeless, state->idx,
                         state->cell_id);
      LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_apply_cells, dim3(grid_for(tot_a)), blk, 0, s, L,
                       (const int64_t *)state->idx, state->cell_id, sh->cell_id_by_id);
    LAUNCH_CHECK();
    if (tot_b > 0) {
      hipLaunchKernelGGL(k_apply_rows, dim3(grid_for(tot_b)), blk, 0, s, L, sh->cell_owned,
                         sh->role, sh->multiplicity, sh->attributes, state->cell_origin,
                         state->position_in_cell);
      LAUNCH_CHECK();
    }
  }
  // {arrivals, the control block, the rainfall} back in one copy
  int64_t *final_words = final10;
  hipLaunchKernelGGL(k_disp_final, one, dim3(16), 0, s, (const int64_t *)(counters + 6),
                     (const int64_t *)state->ctl, (const double *)X.rain, final_words);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, final_words, sizeof(int64_t) * 10, hipMemcpyDeviceToHost,
                         s));
  HIP_TRY(hipStreamSynchronize(s));
  sh->n_arrived = tot_b > 0 ? ctx->mailbox[0] : 0;
  memcpy(rainfall_mass, ctx->mailbox + 9, sizeof(double));
  *valid_n_sd = ctx->mailbox[1];
  if (ctx->mailbox[1 + 7] != 0) {
    sdm_set_error("displacement: grid barrier of the compaction kernel timed out");
    return SDM_E_HIP;
  }
  return SDM_OK;
}
