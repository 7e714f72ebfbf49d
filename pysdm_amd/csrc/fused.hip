// fused.hip -- the per-time-step fused path: one C call = one `Collision.__call__`
// (PySDM/dynamics/collisions/collision.py:174-234), all sub-steps, with the control state
// (lengths, sorted / healthy flags) resident on the device.  Per sub-step:
//   [cell_idx sort by dt_left] -> [counting sort if unsorted] -> PCG64 draws -> shuffle
//   -> k_pair_prob (pairing + sort-within-pair + kernel + probability [+ Ec, fragment mass]
//      [+ per-cell min of optimal dt])
//   -> k_cells (adaptive dt bookkeeping) -> k_pair_update (gamma + multiplicity/attribute update
//      + counters + health flag) -> compaction if unhealthy -> [adaptive end / working length]
#include "common.h"
#include "index.h"
#include "physics.h"

int sdm_adaptive_end_async(sdm_ctx *ctx, const double *dt_left, int64_t n_cell,
                           const int64_t *cell_start, int64_t *scratch2, int64_t *end_dev);

#define CTL_VALID 0
#define CTL_WORK 1
#define CTL_SORTED 2
#define CTL_HEALTHY 3
#define CTL_OVERFLOW 4

#define TID() ((int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x)

struct FusedArgs {
  // state
  int64_t *idx;
  int64_t *multiplicity;
  double *attributes;
  const int64_t *cell_id;
  const int64_t *cell_idx;
  const int64_t *cell_start;
  double *dt_left;
  double *stats_dt_min;
  int64_t *stats_n_substep;
  int64_t *collision_rate, *collision_rate_deficit, *coalescence_rate, *breakup_rate,
      *breakup_rate_deficit;
  const double *gk_a, *gk_b;
  int64_t *ctl;
  // scratch
  const double *rand;     // [P] collision stream
  const double *rand_b;   // [P] proc_rand == rand_frag (same seed, same stream position)
  double *prob;           // [P]
  uint8_t *pair_off;      // [P] 0: pair starts at 2d, 1: at 2d+1, 2: no pair
  int32_t *pair_cid;      // [P] raw cell id of the pair (n_cell > 1)
  double *Ec;             // [P]
  double *fragment_mass;  // [P]
  double *dt_todo, *cell_min, *norm_factor;  // [C]
};

// wave-aggregated int64 counter add: one atomic per wave when all contributing lanes share cid
__device__ __forceinline__ void counter_add(int64_t *__restrict__ counter, int64_t cid,
                                            int64_t v, bool active) {
  active = active && v != 0;
  const unsigned long long am = __ballot(active);
  if (am == 0) return;
  const int first = __ffsll((long long)am) - 1;
  const int64_t cid0 = __shfl((long long)cid, first, 64);
  if (__all(!active || cid == cid0)) {
    const int64_t s = wave_sum_i64(active ? v : 0);
    if (lane_id() == first) atomicAdd((unsigned long long *)&counter[cid0], (unsigned long long)s);
  } else if (active) {
    atomicAdd((unsigned long long *)&counter[cid], (unsigned long long)v);
  }
}

// ---- norm factors (collisions_methods.py:643-650) + per-cell adaptive init -----------------
__global__ void __launch_bounds__(SDM_BLOCK) k_cells_pre(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t c = TID();
  if (c >= cfg.n_cell) return;
  const int64_t sd_num = A.cell_start[c + 1] - A.cell_start[c];
  A.norm_factor[c] = sd_num < 2 ? 0.0
                                : cfg.dt / cfg.dv * (double)sd_num * (double)(sd_num - 1) / 2 /
                                      (double)(sd_num / 2);
  if (cfg.adaptive) {
    const double l = A.dt_left[c];
    A.dt_todo[c] = l < cfg.dt_max ? l : cfg.dt_max;
    A.cell_min[c] = INFINITY;
  }
}

// ---- pairing + probability -------------------------------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK) k_pair_prob(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t W = A.ctl[CTL_WORK];
  const int64_t d = TID();
  const int64_t n_slots = cfg.n_sd / 2;
  bool have = false;
  int64_t i = 0, j = 0, k = 0, cid_j = 0;
  double prob = 0.0, dt_optimal = INFINITY;
  if (d < n_slots) {
    // find_pairs (pair_methods.py:34-55) for positions 2d and 2d+1
    if (cfg.n_cell == 1) {
      if (2 * d + 1 < W) { have = true; i = 2 * d; }
    } else {
#pragma unroll
      for (int o = 0; o < 2 && !have; ++o) {
        const int64_t p = 2 * d + o;
        if (p < W - 1) {
          const int64_t ca = A.cell_id[A.idx[p]], cb = A.cell_id[A.idx[p + 1]];
          const int64_t dd = p - A.cell_start[A.cell_idx[ca]];
          if (ca == cb && (dd & 1) == 0) { have = true; i = p; }
        }
      }
    }
    uint8_t off = 2;
    if (have) {
      off = (uint8_t)(i - 2 * d);
      j = A.idx[i];
      k = A.idx[i + 1];
      int64_t nj = A.multiplicity[j], nk = A.multiplicity[k];
      // sort_within_pair_by_attr (pair_methods.py:126-140)
      if (nj < nk) {
        const int64_t t = j; j = k; k = t;
        const int64_t tn = nj; nj = nk; nk = tn;
        A.idx[i] = j;
        A.idx[i + 1] = k;
      }
      cid_j = cfg.n_cell == 1 ? 0 : A.cell_id[j];
      const double *mass = A.attributes + (int64_t)cfg.mass_attr * cfg.n_sd;
      const double vj = volume_of_mass(mass[j], cfg.rho_w), vk = volume_of_mass(mass[k], cfg.rho_w);
      double rj = 0, rk = 0, uj = 0, uk = 0;
      const bool need_r = cfg.kernel == SDM_KERNEL_GEOMETRIC ||
                          (cfg.enable_breakup && (cfg.ec != SDM_EC_CONST ||
                                                  cfg.frag == SDM_FRAG_STRAUB2010));
      if (need_r) {
        const double inv = 1 / (3.14159265358979323846 * 4 / 3);
        rj = radius_of_volume(vj, inv);
        rk = radius_of_volume(vk, inv);
        if (A.gk_a) {
          uj = gk_interpolate(rj, cfg.gk_factor, A.gk_a, A.gk_b, cfg.gk_table_len);
          uk = gk_interpolate(rk, cfg.gk_factor, A.gk_a, A.gk_b, cfg.gk_table_len);
        }
      }
      double K;
      switch (cfg.kernel) {
        case SDM_KERNEL_GOLOVIN: K = (vj + vk) * cfg.kernel_param[0]; break;
        case SDM_KERNEL_GEOMETRIC: {
          const double s = rj + rk;
          K = (s * s) * cfg.kernel_param[0];
          K *= fabs(uj - uk);
          break;
        }
        default: K = cfg.kernel_param[0];
      }
      // collision.py:249-254: prob = max(n) ; *= K ; normalize (cell of RAW SD #d: quirk)
      prob = (double)nj;
      prob *= K;
      prob *= cfg.n_cell == 1 ? A.norm_factor[0] : A.norm_factor[A.cell_idx[A.cell_id[d]]];
      if (cfg.enable_breakup) {
        double ec;
        switch (cfg.ec) {
          case SDM_EC_CONST: ec = cfg.ec_param[0]; break;
          case SDM_EC_BERRY1967: {
            const double e = linear_collection_efficiency(cfg.berry_params, rj, rk, cfg.berry_unit);
            ec = e * e;
            break;
          }
          default: {  // coalescence_efficiencies/straub2010.py:27-50
            double tmp = vj + vk;
            double Sc = tmp * (6 / 3.14159265358979323846);
            tmp *= 2;
            double tmp2 = fabs(uj - uk);
            tmp2 = tmp2 * tmp2;
            double We = vj * vk;
            if (tmp != 0.0) We /= tmp;
            We *= tmp2;
            We *= cfg.rho_w;
            Sc = signed_pow(Sc, 2.0 / 3.0);
            Sc *= 3.14159265358979323846 * cfg.sgm_w;
            if (Sc != 0.0) We /= Sc;
            We *= -1.15;
            ec = exp(We);
          }
        }
        A.Ec[d] = ec;
        const double u = A.rand_b[d];
        double fm;
        switch (cfg.frag) {
          case SDM_FRAG_ALWAYS_N: fm = (mass[j] + mass[k]) / cfg.frag_param[0]; break;
          case SDM_FRAG_EXPONENTIAL: {
            const double a = 1 - u;
            double fv = -cfg.frag_param[0] * log(a > 1e-5 ? a : 1e-5), nf;
            fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, vj + vk);
            fm = cfg.rho_w * fv;
            break;
          }
          default: {  // breakup_fragmentations/straub2010.py:42-101
            const double v_max = vj > vk ? vj : vk;
            const double x_plus_y = vj + vk;
            const double ds = (rj < rk ? rj : rk) * 2;
            double tmp = vj + vk;
            double Sc = signed_pow(tmp, 2.0 / 3.0);
            Sc *= cfg.frag_param[1];  // PI * sgm_w * (6/PI)**(2/3), one host-side constant
            double tmp2 = fabs(uj - uk);
            tmp2 = tmp2 * tmp2;
            double CKE = vj * vk;
            if (tmp != 0.0) CKE /= tmp;
            CKE *= tmp2;
            CKE *= cfg.rho_w / 2;
            double We = CKE;
            if (Sc != 0.0) We /= Sc;
            double CW = We;
            CW *= CKE;
            CW /= 1e-6;  // si.uJ
            double gam = rj > rk ? rj : rk;
            const double rmin = rj < rk ? rj : rk;
            if (rmin != 0.0) gam /= rmin;
            StraubTmp T = {0, 0, 0, 0, 0, 0};
            double fv = straub_fragment_volume(CW, gam, ds, v_max, u, cfg.straub_consts, T), nf;
            fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, x_plus_y);
            fm = cfg.rho_w * fv;
          }
        }
        A.fragment_mass[d] = fm;
      }
      if (cfg.adaptive && prob != 0) {
        // collisions_methods.py:359-368
        const int64_t prop = nj / nk;
        dt_optimal = cfg.dt * (double)prop / prob;
        dt_optimal = dt_optimal > cfg.dt_min ? dt_optimal : cfg.dt_min;
      }
    }
    A.prob[d] = prob;
    A.pair_off[d] = off;
    if (cfg.n_cell > 1) A.pair_cid[d] = (int32_t)cid_j;
  }
  if (cfg.adaptive) {
    const bool active = have && prob != 0;
    const unsigned long long am = __ballot(active);
    if (am != 0) {
      const int first = __ffsll((long long)am) - 1;
      const int64_t cid0 = __shfl((long long)cid_j, first, 64);
      if (__all(!active || cid_j == cid0)) {
        const double m = wave_min_f64(active ? dt_optimal : INFINITY);
        if (lane_id() == first) atomic_min_pos_f64(&A.cell_min[cid0], m);
      } else if (active) {
        atomic_min_pos_f64(&A.cell_min[cid_j], dt_optimal);
      }
    }
  }
}

// ---- per-cell adaptive bookkeeping (collisions_methods.py:357-374) ---------------------------
__global__ void __launch_bounds__(SDM_BLOCK) k_cells_adaptive(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t c = TID();
  if (c >= cfg.n_cell) return;
  if (A.ctl[CTL_WORK] == 0) return;
  const double m = A.cell_min[c];
  double t = A.dt_todo[c];
  if (m < t) t = m;
  A.dt_todo[c] = t;
  const double s = A.stats_dt_min[c];
  A.stats_dt_min[c] = m < s ? m : s;  // Python min(s, m): NaN-sticky
  A.dt_left[c] -= t;
  if (t > 0) A.stats_n_substep[c] += 1;
}

// ---- gamma + update ----------------------------------------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK) k_pair_update(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t W = A.ctl[CTL_WORK];
  const int64_t d = TID();
  bool collide = false;
  int64_t j = 0, k = 0, cid = 0, nk = 0, gi = 0, gc = 0;
  double g = 0;
  if (d < W / 2) {
    double p = A.prob[d];
    if (p != 0) {
      if (cfg.adaptive) {
        cid = cfg.n_cell > 1 ? A.pair_cid[d] : 0;
        p *= A.dt_todo[cid] / cfg.dt;  // collisions_methods.py:369-372
      } else {
        p /= (double)cfg.substeps;  // collision.py:279
      }
    }
    g = ceil(p - A.rand[d]);  // collisions_methods.py:560
    // pair_indices' skip (gamma == 0) also covers "no pair": off == 2 implies prob == 0,
    // hence gamma = ceil(-rand) = -0.0
    if (g != 0) {
      const int64_t off = A.pair_off[d];
      if (off < 2) {
        collide = true;
        j = A.idx[2 * d + off];
        k = A.idx[2 * d + 1 + off];
        nk = A.multiplicity[k];
        const int64_t prop = A.multiplicity[j] / nk;
        gi = (int64_t)g;
        gc = gi < prop ? gi : prop;
        cid = cfg.n_cell == 1 ? 0 : A.cell_id[j];
        g = (double)gc;
      }
    }
  }
  counter_add(A.collision_rate, cid, gc * nk, collide);
  counter_add(A.collision_rate_deficit, cid, (gi - gc) * nk, collide);
  collide = collide && g != 0;
  bool coal = collide;
  bool ovf = false;
  if (cfg.enable_breakup && collide) {
    const double r = A.rand_b[d], ec = A.Ec[d], eb = cfg.eb_const;
    if (r - (ec + (1 - ec) * eb) > 0) {
      collide = false;  // bounce
      coal = false;
    } else if (!(r - ec < 0)) {
      coal = false;
      const double *mass = A.attributes + (int64_t)cfg.mass_attr * cfg.n_sd;
      // break_up / break_up_while (collisions_methods.py:135-243); counters through atomics
      double gamma_deficit = g;
      if (!cfg.handle_all_breakups) {
        double take_from_j, new_mult_k;
        int64_t gamma_j_k;
        compute_transfer_multiplicities(g, A.multiplicity[j], nk, mass[j], mass[k],
                                        A.fragment_mass[d], cfg.max_multiplicity, take_from_j,
                                        new_mult_k, gamma_j_k, ovf);
        gamma_deficit = g - (double)gamma_j_k;
        if (gamma_j_k) atomicAdd((unsigned long long *)&A.breakup_rate[cid],
                                 (unsigned long long)(gamma_j_k * nk));
        if (gamma_deficit != 0)
          atomicAdd((unsigned long long *)&A.breakup_rate_deficit[cid],
                    (unsigned long long)(int64_t)(gamma_deficit * (double)nk));
        apply_breakup_transfer(j, k, take_from_j, new_mult_k, A.multiplicity, A.attributes,
                               cfg.n_attr, cfg.n_sd);
      } else {
        const double fm = A.fragment_mass[d];
        while (gamma_deficit > 0) {
          double take_from_j, new_mult_k, gamma_j_k;
          const int64_t mj = A.multiplicity[j], mk = A.multiplicity[k];
          if (mk == mj) {
            take_from_j = (double)mj;
            new_mult_k = (mass[j] + mass[k]) / fm * (double)mk;
            if (new_mult_k > (double)cfg.max_multiplicity) {
              atomicAdd((unsigned long long *)&A.breakup_rate_deficit[cid],
                        (unsigned long long)(int64_t)(gamma_deficit * (double)mk));
              ovf = true;
              break;
            }
            gamma_j_k = gamma_deficit;
          } else {
            if (mk > mj) { const int64_t t = j; j = k; k = t; }
            int64_t g_int;
            compute_transfer_multiplicities(gamma_deficit, A.multiplicity[j], A.multiplicity[k],
                                            mass[j], mass[k], fm, cfg.max_multiplicity,
                                            take_from_j, new_mult_k, g_int, ovf);
            gamma_j_k = (double)g_int;
          }
          const int64_t add = (int64_t)(gamma_j_k * (double)A.multiplicity[k]);
          if (add) atomicAdd((unsigned long long *)&A.breakup_rate[cid], (unsigned long long)add);
          gamma_deficit -= gamma_j_k;
          apply_breakup_transfer(j, k, take_from_j, new_mult_k, A.multiplicity, A.attributes,
                                 cfg.n_attr, cfg.n_sd);
        }
        const int64_t add = (int64_t)(gamma_deficit * (double)A.multiplicity[k]);
        if (add) atomicAdd((unsigned long long *)&A.breakup_rate_deficit[cid],
                           (unsigned long long)add);
      }
      if (ovf) atomicAdd((unsigned long long *)&A.ctl[CTL_OVERFLOW], 1ull);
    }
  }
  counter_add(A.coalescence_rate, cid, (int64_t)(g * (double)nk), coal);
  if (coal) coalesce_pair(j, k, g, A.multiplicity, A.attributes, cfg.n_attr, cfg.n_sd);
  if (collide && (A.multiplicity[k] == 0 || A.multiplicity[j] == 0)) A.ctl[CTL_HEALTHY] = 0;
}

// ---- control-word kernels -------------------------------------------------------------------
// conditional counting sort support: the sort cores read the length from ctl[CTL_WORK]; whether
// to run is decided on the device by ctl[CTL_SORTED].
__global__ void k_sort_gate(int64_t *ctl, int64_t *gate_len) {
  // gate_len[0] = length to sort (0 disables every kernel of the sort core)
  gate_len[0] = ctl[CTL_SORTED] ? 0 : ctl[CTL_WORK];
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_sort_commit(int64_t *__restrict__ idx, const int64_t *__restrict__ sorted_buf,
              int64_t *__restrict__ cell_start, const int64_t *__restrict__ cs_tmp,
              int64_t n_cell, int64_t *ctl, const int64_t *__restrict__ gate_len) {
  const int64_t n = gate_len[0];
  if (ctl[CTL_SORTED]) return;
  const int64_t i = TID();
  if (i < n) idx[i] = sorted_buf[i];
  if (i <= n_cell) cell_start[i] = cs_tmp[i];
}

__global__ void k_sort_done(int64_t *ctl) { ctl[CTL_SORTED] = 1; }
__global__ void k_mark_unsorted(int64_t *ctl) { ctl[CTL_SORTED] = 0; }

__global__ void k_post_sanitize(int64_t *ctl, const int64_t *cctl) {
  if (ctl[CTL_HEALTHY] == 0) {
    ctl[CTL_VALID] = cctl[1];
    ctl[CTL_WORK] = cctl[1];
    ctl[CTL_SORTED] = 0;
    ctl[CTL_HEALTHY] = 1;
  }
}

__global__ void k_pre_sanitize(int64_t *ctl) {
  // particle_attributes.py:69: idx.length = valid_n_sd before removal
  if (ctl[CTL_HEALTHY] == 0) ctl[CTL_WORK] = ctl[CTL_VALID];
}

__global__ void k_set_work(int64_t *ctl, const int64_t *end) { ctl[CTL_WORK] = end[0]; }
__global__ void k_reset_work(int64_t *ctl) { ctl[CTL_WORK] = ctl[CTL_VALID]; }

__global__ void __launch_bounds__(SDM_BLOCK)
k_copy_tail(int64_t *__restrict__ dst, const int64_t *__restrict__ src,
            const int64_t *__restrict__ p_from, int64_t n) {
  const int64_t i = TID();
  if (i >= *p_from && i < n) dst[i] = src[i];
}

__global__ void __launch_bounds__(SDM_BLOCK) k_fill_f64(double *p, double v, int64_t n) {
  const int64_t i = TID();
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
struct FusedScratch {
  double *pairs_rand, *rand, *rand_b, *prob, *Ec, *fragment_mass, *dt_todo, *cell_min,
      *norm_factor;
  uint8_t *pair_off;
  int32_t *pair_cid;
  int64_t *sorted_buf, *cs_tmp, *gate_len, *cctl, *end2;
  char *shuffle, *sort, *compact;
  size_t total;
};

static FusedScratch layout(char *base, const sdm_step_cfg *cfg, int64_t shift) {
  FusedScratch S;
  Carver cv(base);
  const int64_t N = cfg->n_sd, P = N / 2 > 0 ? N / 2 : 1, C = cfg->n_cell;
  S.pairs_rand = cv.take<double>(N + shift);
  S.rand = cv.take<double>(P);
  S.rand_b = cv.take<double>(P);
  S.prob = cv.take<double>(P);
  S.Ec = cv.take<double>(P);
  S.fragment_mass = cv.take<double>(P);
  S.dt_todo = cv.take<double>(C);
  S.cell_min = cv.take<double>(C);
  S.norm_factor = cv.take<double>(C);
  S.pair_off = cv.take<uint8_t>(P);
  S.pair_cid = cv.take<int32_t>(P);
  S.sorted_buf = cv.take<int64_t>(N);
  S.cs_tmp = cv.take<int64_t>(C + 1);
  S.gate_len = cv.take<int64_t>(4);
  S.cctl = cv.take<int64_t>(8);
  S.end2 = cv.take<int64_t>(4);
  S.shuffle = base + cv.off;
  cv.off += carve_size(sdm_shuffle_scratch(N));
  S.sort = base + cv.off;
  cv.off += carve_size(sdm_sort_scratch(N, C));
  S.compact = base + cv.off;
  cv.off += carve_size(sdm_compact_scratch(N));
  S.total = cv.off;
  return S;
}

static int cond_sort(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st,
                     const FusedScratch &S) {
  hipLaunchKernelGGL(k_sort_gate, dim3(1), dim3(1), 0, ctx->stream, st->ctl, S.gate_len);
  LAUNCH_CHECK();
  int rc = sdm_counting_sort_async(ctx, S.sort, S.sorted_buf, st->idx, st->cell_id, st->cell_idx,
                                   S.gate_len, cfg->n_sd, S.cs_tmp, cfg->n_cell);
  if (rc) return rc;
  const int64_t n = cfg->n_sd > cfg->n_cell + 1 ? cfg->n_sd : cfg->n_cell + 1;
  hipLaunchKernelGGL(k_sort_commit, dim3(grid_for(n)), dim3(SDM_BLOCK), 0, ctx->stream, st->idx,
                     S.sorted_buf, st->cell_start, S.cs_tmp, cfg->n_cell, st->ctl, S.gate_len);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sort_done, dim3(1), dim3(1), 0, ctx->stream, st->ctl);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_collision_step(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st,
                                  sdm_step_result *res, int read_back) {
  ARG_TRY(ctx && cfg && st && res);
  ARG_TRY(cfg->n_sd >= 2 && cfg->n_sd < INT32_MAX && cfg->n_cell >= 1 && cfg->n_attr >= 1);
  ARG_TRY(st->idx && st->tmp_idx && st->multiplicity && st->attributes && st->cell_id &&
          st->cell_idx && st->cell_start && st->dt_left && st->stats_dt_min &&
          st->stats_n_substep && st->collision_rate && st->collision_rate_deficit &&
          st->coalescence_rate && st->ctl);
  ARG_TRY(!cfg->enable_breakup || (st->breakup_rate && st->breakup_rate_deficit));
  ARG_TRY(cfg->kernel != SDM_KERNEL_GEOMETRIC || (st->gk_a && st->gk_b && cfg->gk_table_len > 0));
  ARG_TRY(cfg->adaptive || cfg->substeps >= 1);
  ARG_TRY(cfg->mass_attr >= 0 && cfg->mass_attr < cfg->n_attr);
  ARG_TRY(cfg->dt_min > 0);

  const int64_t N = cfg->n_sd, P = N / 2, C = cfg->n_cell;
  // random_generator_optimizer.py:21-25
  int64_t shift = 0;
  if (cfg->optimized_random) {
    const double q = cfg->dt / cfg->dt_min;
    shift = (int64_t)q;
    if ((double)shift < q) shift += 1;
  }
  FusedScratch S = layout(nullptr, cfg, shift);
  int rc = sdm_reserve(ctx, S.total);
  if (rc) return rc;
  S = layout(ctx->arena, cfg, shift);

  FusedArgs A;
  memset(&A, 0, sizeof(A));
  A.multiplicity = st->multiplicity;
  A.attributes = st->attributes;
  A.cell_id = st->cell_id;
  A.cell_idx = st->cell_idx;
  A.cell_start = st->cell_start;
  A.dt_left = st->dt_left;
  A.stats_dt_min = st->stats_dt_min;
  A.stats_n_substep = st->stats_n_substep;
  A.collision_rate = st->collision_rate;
  A.collision_rate_deficit = st->collision_rate_deficit;
  A.coalescence_rate = st->coalescence_rate;
  A.breakup_rate = st->breakup_rate;
  A.breakup_rate_deficit = st->breakup_rate_deficit;
  A.gk_a = st->gk_a;
  A.gk_b = st->gk_b;
  A.ctl = st->ctl;
  A.rand = S.rand;
  A.rand_b = S.rand_b;
  A.prob = S.prob;
  A.pair_off = S.pair_off;
  A.pair_cid = S.pair_cid;
  A.Ec = S.Ec;
  A.fragment_mass = S.fragment_mass;
  A.dt_todo = S.dt_todo;
  A.cell_min = S.cell_min;
  A.norm_factor = S.norm_factor;

  uint64_t off = st->rng_offset, off_b = st->rng_offset_breakup;
  int64_t n_sub = 0, n_pairs = 0, swaps = 0;
  int64_t *cur = st->idx, *alt = st->tmp_idx;
  hipStream_t s = ctx->stream;
  const dim3 blk(SDM_BLOCK);

  if (cfg->adaptive) {  // collision.py:180: dt_left[:] = dt
    hipLaunchKernelGGL(k_fill_f64, dim3(grid_for(C)), blk, 0, s, st->dt_left, cfg->dt, C);
    LAUNCH_CHECK();
  }
  int64_t work_host = -1;
  if (cfg->adaptive || read_back) {
    HIP_TRY(hipMemcpyAsync(ctx->mailbox + 3, st->ctl + CTL_WORK, sizeof(int64_t),
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    work_host = ctx->mailbox[3];
  }
  for (;;) {
    if (!cfg->adaptive && n_sub >= cfg->substeps) break;
    if (cfg->adaptive && work_host == 0) break;
    // (a) collision.py:183 cell_idx.sort_by_key(dt_left)
    if (cfg->adaptive && C > 1) {
      rc = sdm_sort_by_key_async(ctx, st->cell_idx, st->dt_left, C);
      if (rc) return rc;
    }
    // (b) cell_start getter: counting sort if unsorted (particle_attributes.py:51-55)
    sdm_step_state cur_state = *st;
    cur_state.idx = cur;
    {
      PhaseScope ph(ctx, SDM_PHASE_SORT);
      rc = cond_sort(ctx, cfg, &cur_state, S);
      if (rc) return rc;
    }
    // (c) random numbers (random_generator_optimizer.py:37-48)
    const double *u01;
    if (!cfg->optimized_random || n_sub == 0) {
      PhaseScope ph(ctx, SDM_PHASE_RNG);
      rc = sdm_pcg_fill_async(ctx, S.pairs_rand, N + shift, cfg->rng_state_inc, off);
      if (rc) return rc;
      rc = sdm_pcg_fill_async(ctx, S.rand, P, cfg->rng_state_inc, off + (uint64_t)(N + shift));
      if (rc) return rc;
      off += (uint64_t)(N + shift + P);
      if (cfg->enable_breakup) {
        rc = sdm_pcg_fill_async(ctx, S.rand_b, P, cfg->rng_state_inc, off_b);
        if (rc) return rc;
        off_b += (uint64_t)P;
      }
    }
    u01 = S.pairs_rand + (cfg->optimized_random ? n_sub : 0);
    // (d) permutation (particle_attributes.py:98-105)
    // shuffle_local visits every cell of cell_start, also those beyond a cut working length
    // (index_methods.py:35); shuffle_global covers the working length (index.py:43-45)
    const int64_t *p_shuffle_len = cfg->croupier_local ? st->cell_start + C : st->ctl + CTL_WORK;
    rc = sdm_shuffle_async(ctx, S.shuffle, alt, cur, u01, st->cell_start, C, p_shuffle_len, N,
                           !cfg->croupier_local);
    if (rc) return rc;
    // positions beyond the working length keep their content: the shuffle core writes only
    // [0, work) of `alt`, so carry the rest over
    {
      PhaseScope ph(ctx, SDM_PHASE_TAIL_COPY);
      hipLaunchKernelGGL(k_copy_tail, dim3(grid_for(N)), blk, 0, s, alt, cur, p_shuffle_len, N);
      LAUNCH_CHECK();
    }
    { int64_t *t = cur; cur = alt; alt = t; }
    ++swaps;
    if (!cfg->croupier_local) {
      hipLaunchKernelGGL(k_mark_unsorted, dim3(1), dim3(1), 0, s, st->ctl);
      LAUNCH_CHECK();
      cur_state.idx = cur;
      rc = cond_sort(ctx, cfg, &cur_state, S);
      if (rc) return rc;
    }
    A.idx = cur;
    // (e) probabilities
    {
      PhaseScope ph(ctx, SDM_PHASE_CELLS_PRE);
      hipLaunchKernelGGL(k_cells_pre, dim3(grid_for(C)), blk, 0, s, *cfg, A);
      LAUNCH_CHECK();
    }
    {
      PhaseScope ph(ctx, SDM_PHASE_PAIR_PROB);
      hipLaunchKernelGGL(k_pair_prob, dim3(grid_for(P)), blk, 0, s, *cfg, A);
      LAUNCH_CHECK();
    }
    if (cfg->adaptive) {
      PhaseScope ph(ctx, SDM_PHASE_CELLS_ADAPTIVE);
      hipLaunchKernelGGL(k_cells_adaptive, dim3(grid_for(C)), blk, 0, s, *cfg, A);
      LAUNCH_CHECK();
    }
    // (f) gamma + update
    {
      PhaseScope ph(ctx, SDM_PHASE_PAIR_UPDATE);
      hipLaunchKernelGGL(k_pair_update, dim3(grid_for(P)), blk, 0, s, *cfg, A);
      LAUNCH_CHECK();
    }
    // (g) sanitize (particle_attributes.py:67-73)
    {
      PhaseScope ph(ctx, SDM_PHASE_SANITIZE);
      hipLaunchKernelGGL(k_pre_sanitize, dim3(1), dim3(1), 0, s, st->ctl);
      LAUNCH_CHECK();
      rc = sdm_compact_async(ctx, S.compact, st->multiplicity, cur, st->ctl + CTL_WORK, N, N,
                             st->ctl + CTL_HEALTHY, S.cctl);
      if (rc) return rc;
      hipLaunchKernelGGL(k_post_sanitize, dim3(1), dim3(1), 0, s, st->ctl, S.cctl);
      LAUNCH_CHECK();
    }
    ++n_sub;
    if (!cfg->adaptive && work_host >= 0) n_pairs += work_host / 2;
    if (cfg->adaptive) {
      // (h) collision.py:185-187 cut_working_length(adaptive_sdm_end(dt_left))
      cur_state.idx = cur;
      rc = cond_sort(ctx, cfg, &cur_state, S);
      if (rc) return rc;
      n_pairs += work_host / 2;
      rc = sdm_adaptive_end_async(ctx, st->dt_left, C, st->cell_start, S.end2, S.end2 + 1);
      if (rc) return rc;
      hipLaunchKernelGGL(k_set_work, dim3(1), dim3(1), 0, s, st->ctl, S.end2 + 1);
      LAUNCH_CHECK();
      HIP_TRY(hipMemcpyAsync(ctx->mailbox + 3, st->ctl + CTL_WORK, sizeof(int64_t),
                             hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      work_host = ctx->mailbox[3];
    }
  }
  if (cfg->adaptive) {
    // collision.py:189-190 reset_working_length(); reset_cell_idx() (identity + sort)
    hipLaunchKernelGGL(k_reset_work, dim3(1), dim3(1), 0, s, st->ctl);
    LAUNCH_CHECK();
    if (C > 1) {
      rc = sdm_identity_index(ctx, st->cell_idx, C);
      if (rc) return rc;
      hipLaunchKernelGGL(k_mark_unsorted, dim3(1), dim3(1), 0, s, st->ctl);
      LAUNCH_CHECK();
      sdm_step_state cur_state = *st;
      cur_state.idx = cur;
      rc = cond_sort(ctx, cfg, &cur_state, S);
      if (rc) return rc;
    }
  }
  res->n_substeps = n_sub;
  res->idx_swapped = swaps & 1;
  res->rng_offset = off;
  res->rng_offset_breakup = off_b;
  res->valid_n_sd = -1;
  if (read_back) {
    HIP_TRY(hipMemcpyAsync(ctx->mailbox, st->ctl, sizeof(int64_t) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    res->valid_n_sd = ctx->mailbox[CTL_VALID];
  }
  // non-adaptive without read-back: the caller derives the pair count from its own length
  res->n_pairs = (cfg->adaptive || work_host >= 0) ? n_pairs : -1;
  st->rng_offset = off;
  st->rng_offset_breakup = off_b;
  return SDM_OK;
}
