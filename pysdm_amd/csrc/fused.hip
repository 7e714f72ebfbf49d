// fused.hip -- the per-time-step fused path: one C call = one `Collision.__call__`
// (PySDM/dynamics/collisions/collision.py:174-234), all sub-steps, with the control state
// (lengths, sorted / healthy flags) resident on the device.  Three routes per sub-step:
//   one cell, non-adaptive:  shuffle record build (index.hip) -> k_pair_all (walks, pairing,
//      sort-within-pair, kernel, probability, gamma, update, counters) -> gated compaction;
//      inside a run of several steps: k_bin_build2 -> k_pair_all_sort, two launches per step - the
//      pair kernel carries the tile sort of the next step's build, the build the compaction (and
//      the repeated sort) after the rare step in which a super-droplet died
//   one cell, adaptive:  build -> k_pair_prob -> [k_cells_adaptive: above 2048 partial minima;
//      else folded into] k_pair_update (the per-cell minimum of the optimal sub-step is the one
//      global dependency) [-> k_resolve_dense with breakup] -> compaction (from the list of the
//      dead that the kernels before it kept), whose epilogue closes the sub-step and publishes the
//      control block; the head of the next sub-step (build + k_pair_prob) is launched ahead of
//      the read-back
//   many cells of at most 6144 super-droplets:  k_cells_turn (ends the previous sub-step, opens
//      this one) -> k_cell_step2 / k_cell_step (one workgroup per cell: shuffle in LDS, pairs,
//      probabilities, update) [-> all-reduce MIN in sharded runs]; the next sub-step is launched
//      ahead, gated on the device; a compaction only when the host has learnt of a death
//   larger cells / global croupier:  generic kernels (per-position cell look-ups, counting sort)
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "index.h"
#include "shuffle_build.h"
#include "physics.h"

int sdm_adaptive_end_async(sdm_ctx *ctx, const double *dt_left, int64_t n_cell,
                           const int64_t *cell_start, int64_t *scratch2, int64_t *end_dev);

#define CTL_VALID 0
#define CTL_WORK 1
#define CTL_SORTED 2
#define CTL_HEALTHY 3
#define CTL_OVERFLOW 4
#define CTL_PAIRS 5  // candidate pairs processed by the single-cell non-adaptive pair kernel

#define TID() ((int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x)

// per-super-droplet mirror record: {multiplicity, mass} (16 B) or, when radius / terminal velocity
// are needed (geometric kernel, Berry / Straub breakup), {multiplicity, mass, radius, velocity}
// (32 B): the derived attributes are then evaluated once per change of a droplet instead of once
// per candidate pair and sub-step (attributes/physics/{radius,terminal_velocity}.py semantics)
struct SD { int64_t n; double m, r, u; };
struct Collided;

struct FusedArgs {
  // state
  int64_t *idx;
  int64_t *multiplicity;
  double *attributes;
  const int64_t *cell_id;
  const int64_t *cell_id_raw;  // the caller's cell_id column (== cell_id unless a run relabelled)
  const int64_t *cell_idx;
  const int64_t *cell_start;
  double *dt_left;
  double *stats_dt_min;
  int64_t *stats_n_substep;
  int64_t *collision_rate, *collision_rate_deficit, *coalescence_rate, *breakup_rate,
      *breakup_rate_deficit;
  const double *gk_a, *gk_b;
  int64_t *ctl;
  // sharded mode (sdm_hip.h): by cell id, 1 = this process computes the cell; NULL = all
  const uint8_t *cell_owned;
  // ... the same by SEGMENT of the permutation (1 = the segment's cell is this process's), written
  // by k_cells_turn for the cell kernel it opens: a workgroup that has nothing to do learns so
  // from one load instead of a chain of four (cell_start -> permutation -> cell id -> mask).  NULL:
  // not available
  const uint8_t *seg_owned;
  // ... and the cell id of every segment (adaptive per-cell route, sharded or not); NULL: unknown
  const int32_t *seg_cid;
  // sharded mode: super-droplets that died in this process's cells in the current sub-step, counted
  // where they are flagged (the list of their positions is built only once the exchange of the
  // counts has shown that somebody's did: k_shard_dead_list); NULL otherwise
  unsigned long long *n_dead;
  // adaptive steps of one cell: the positions of the dead are listed as they are flagged (at most
  // SDM_DEAD_LIST_CAP are kept; the count goes on), for a compaction without a pass over the
  // permutation (index.hip: compact_listed_body); NULL otherwise
  int64_t *dead_pos;
  unsigned long long *dead_count;
  // sharded adaptive per-cell route: minus the deaths per SEGMENT of the permutation (k_cells_turn's
  // min_out + n_cell: completed over the processes by the same all-reduce MIN as the cell minima)
  double *seg_deaths;
  // graph replay (common.h: gwords): s_rand / s_rand_b are then the generators' initial states and
  // the stream positions come from the device ({collision stream, breakup streams}); rand_extra =
  // distance from a draw's first u01 to its first `rand` (n_sd + shift)
  const uint64_t *dev_off;
  uint64_t rand_extra;
  // scratch
  // PCG64 streams, evaluated in the kernels (no u01 arrays): `s_rand` = state of the collision
  // generator at the first draw of `rand` (after pairs_rand), `s_rand_b` = state of the
  // proc_rand / rand_frag generators (same seed, same position: identical values)
  u128 s_rand, s_rand_b, rng_inc;
  const u128 *rng_tab;
  const u128 *rng_aff;  // ready jumps (common.h: pcg_aff); NULL while graph replay is captured
  double *prob;           // [P]
  uint8_t *pair_off;      // [P] 0: pair starts at 2d, 1: at 2d+1, 2: no pair
  int32_t *pair_cid;      // [P] raw cell id of the pair (n_cell > 1)
  double *dt_todo, *cell_min;  // [C]
  double *block_min;           // single cell: per-workgroup partial minima of the optimal dt
  int n_block_min;
  // single adaptive cell, at most 2048 partial minima: k_pair_update does k_cells_adaptive's work
  // itself (0: no; 1: yes; 2: yes, and this is the first sub-step of the time step).  dt_left_pub:
  // dt_left[0] as the previous sub-step left it, in a word nobody writes during this kernel (the
  // compaction's epilogue copies it there)
  int fold_pre;
  const double *dt_left_pub;
  // single-cell fast path: the pair kernels do the shuffle's backward walk themselves (two
  // positions per thread) and write the permuted, pair-sorted idx once; NULL otherwise
  const void *rec;  // records of layout rec_fmt (shuffle_device.h: SDM_REC_*)
  int rec_fmt;
  const int32_t *ovf_head, *ovf_next;
  const int64_t *idx_prev;  // previous permutation (source of the dead tail)
  // {multiplicity, mass} of each super-droplet side by side (one random line per gather instead
  // of two); a mirror of the SoA columns kept current by the update code, NULL = not in use
  double *nm;
  int nm_wide;  // 0: 16-B records, 1: 32-B records
  // one cell: every wave of a step would add to the same few counter words, and same-address
  // atomics serialise in L2 (~4 ns each: 70 us per step once most waves hold a colliding pair).
  // The adds go to slot (workgroup % SDM_CNT_SLOTS) instead - one cache line per slot - and
  // k_fold_counters adds the slots to the counters at the end of the call.  NULL: n_cell > 1
  int64_t *slots;
  // breakup: colliding pairs listed by the pair kernels, resolved by k_resolve_dense
  // `list_nl` lists of capacity `list_cap` each (list l starts at list + l * list_cap, its fill
  // count is list_count[l * SDM_CNT_STRIDE]); a workgroup appends to list (blockIdx % list_nl), so
  // the appends of a launch do not all queue up on one counter word
  struct Collided *list;
  unsigned long long *list_count;
  // the fill counts are double-buffered by sub-step: k_resolve_dense clears the set the next
  // sub-step will use (no memset launch per sub-step)
  unsigned long long *list_count_next;
  int list_nl;
  int64_t list_cap;
};

#define LIST_NL 64  // lists of colliding pairs (flat pair kernels; the per-cell kernel uses one)
enum { CNT_COLLISION = 0, CNT_COLLISION_DEFICIT, CNT_COALESCENCE, CNT_BREAKUP, CNT_BREAKUP_DEFICIT,
       CNT_OVERFLOW, CNT_KINDS };
static_assert(CNT_OVERFLOW == SDM_CNT_OVERFLOW && CTL_OVERFLOW == 4,
              "index.hip:compact_epilogue folds this slot word into control word 4");

__device__ __forceinline__ int64_t *counter_of(const FusedArgs &A, int which) {
  return which == CNT_COLLISION ? A.collision_rate
         : which == CNT_COLLISION_DEFICIT ? A.collision_rate_deficit
         : which == CNT_COALESCENCE ? A.coalescence_rate
         : which == CNT_BREAKUP ? A.breakup_rate
         : which == CNT_BREAKUP_DEFICIT ? A.breakup_rate_deficit : A.ctl + CTL_OVERFLOW;
}

// wave-aggregated int64 counter add: one atomic per wave when all contributing lanes share cid.
// Wave-collective: lanes that have nothing to add call it with active = false.
__device__ __forceinline__ void counter_add(const FusedArgs &A, int which, int64_t cid, int64_t v,
                                            bool active) {
  active = active && v != 0;
  const unsigned long long am = __ballot(active);
  if (am == 0) return;
  const int first = __ffsll((long long)am) - 1;
  if (A.slots) {  // one cell
    const int64_t s = wave_sum_i64(active ? v : 0);
    if (lane_id() == first)
      atomicAdd((unsigned long long *)&A.slots[(blockIdx.x & (SDM_CNT_SLOTS - 1)) * SDM_CNT_STRIDE +
                                               which], (unsigned long long)s);
    return;
  }
  wave_counter_add(counter_of(A, which), cid, v, active);
}

// adds the slots to the (single) cell's counters and clears them; one workgroup of SDM_CNT_SLOTS
__global__ void __launch_bounds__(SDM_CNT_SLOTS) k_fold_counters(FusedArgs A) {
  __shared__ int64_t part[SDM_CNT_SLOTS / SDM_WAVE][CNT_KINDS];
  int64_t *slot = A.slots + threadIdx.x * SDM_CNT_STRIDE;
  for (int w = 0; w < CNT_KINDS; ++w) {
    const int64_t v = slot[w];
    if (v != 0) slot[w] = 0;
    const int64_t s = wave_sum_i64(v);
    if (lane_id() == 0) part[threadIdx.x / SDM_WAVE][w] = s;
  }
  __syncthreads();
  if (threadIdx.x < CNT_KINDS) {
    int64_t s = 0;
    for (int w = 0; w < SDM_CNT_SLOTS / SDM_WAVE; ++w) s += part[w][threadIdx.x];
    if (s != 0) counter_of(A, threadIdx.x)[0] += s;
  }
}

// one draw per thread: element (block_first + tid) of the stream starting at `s_base`
__device__ __forceinline__ double stream_draw(u128 s_base, u128 inc, const u128 *__restrict__ tab,
                                              u128 *lds_slot, uint64_t add = 0,
                                              const u128 *__restrict__ aff = nullptr) {
  // ready affine maps (common.h: pcg_aff): block b starts b * SDM_BLOCK draws in =
  // (b / PER) strides + (b % PER) * SDM_BLOCK, so two multiply-adds per thread and no exchange
  // through LDS, instead of a bit-by-bit jump by thread 0 and another by every thread
  constexpr int PER = PCG_AFF_STRIDE / SDM_BLOCK;
  if (aff && add == 0 && blockIdx.x / PER < PCG_AFF_TILES) {
    u128 state = pcg_apply(pcg_apply(s_base, aff, PCG_AFF_SMALL + (int64_t)(blockIdx.x / PER)),
                           aff, (int64_t)(blockIdx.x % PER) * SDM_BLOCK + threadIdx.x);
    state = state * pcg_mult() + inc;
    return pcg_output(state);
  }
  if (threadIdx.x == 0) *lds_slot = pcg_jump(s_base, tab, (uint64_t)blockIdx.x * SDM_BLOCK + add);
  __syncthreads();
  u128 state = pcg_jump(*lds_slot, tab, (uint64_t)threadIdx.x);
  state = state * pcg_mult() + inc;
  return pcg_output(state);
}

// collisions_methods.py:643-650 (left-to-right evaluation)
__device__ __forceinline__ double norm_factor_of(const sdm_step_cfg &cfg,
                                                 const int64_t *__restrict__ cell_start,
                                                 int64_t c) {
  const int64_t sd_num = cell_start[c + 1] - cell_start[c];
  return sd_num < 2 ? 0.0
                    : cfg.dt / cfg.dv * (double)sd_num * (double)(sd_num - 1) / 2 /
                          (double)(sd_num / 2);
}

__device__ __forceinline__ void derive_ru(const sdm_step_cfg &cfg, const FusedArgs &A, SD &v) {
  const double inv = 1 / (3.14159265358979323846 * 4 / 3);
  v.r = radius_of_volume(volume_of_mass(v.m, cfg.rho_w), inv);
  v.u = A.gk_a ? gk_interpolate(v.r, cfg.gk_factor, A.gk_a, A.gk_b, cfg.gk_table_len) : 0.0;
}

// state of super-droplet `id`: from the mirror if there is one, else from the SoA columns
__device__ __forceinline__ SD sd_load(const sdm_step_cfg &cfg, const FusedArgs &A, int64_t id,
                                      bool need_ru) {
  SD v;
  if (A.nm && A.nm_wide) {
    const double4 w = ((const double4 *)A.nm)[id];
    v.n = __double_as_longlong(w.x); v.m = w.y; v.r = w.z; v.u = w.w;
  } else {
    if (A.nm) {
      const double2 w = ((const double2 *)A.nm)[id];
      v.n = __double_as_longlong(w.x); v.m = w.y;
    } else {
      v.n = A.multiplicity[id];
      v.m = (A.attributes + (int64_t)cfg.mass_attr * cfg.n_sd)[id];
    }
    v.r = v.u = 0.0;
    if (need_ru) derive_ru(cfg, A, v);
  }
  return v;
}

// refresh the mirror record of `id` from the SoA columns (after an update)
__device__ __forceinline__ void sd_refresh(const sdm_step_cfg &cfg, const FusedArgs &A,
                                           int64_t id) {
  if (!A.nm) return;
  SD v;
  v.n = A.multiplicity[id];
  v.m = (A.attributes + (int64_t)cfg.mass_attr * cfg.n_sd)[id];
  if (A.nm_wide) {
    derive_ru(cfg, A, v);
    ((double4 *)A.nm)[id] = make_double4(__longlong_as_double(v.n), v.m, v.r, v.u);
  } else {
    ((double2 *)A.nm)[id] = make_double2(__longlong_as_double(v.n), v.m);
  }
}

// ---- per-cell adaptive init (collisions_methods.py:355-356) -----------------------------------
__global__ void __launch_bounds__(SDM_BLOCK) k_cells_pre(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t c = TID();
  if (c >= cfg.n_cell) return;
  const double l = A.dt_left[c];
  A.dt_todo[c] = cfg.dt_max < l ? cfg.dt_max : l;  // Python min(l, dt_max)
  A.cell_min[c] = INFINITY;
}

// Ec (coalescence efficiency) and fragment mass of one pair from the members' masses -- only
// evaluated for pairs that actually collide (they are pure functions of the pair's state before
// the update, so evaluating them lazily gives the values the reference computes for all pairs)
__device__ __forceinline__ void breakup_params(const sdm_step_cfg &cfg, const FusedArgs &A,
                                               const SD &sj, const SD &sk, double u_b,
                                               double &ec, double &fm) {
  const double mj = sj.m, mk = sk.m;
  const double vj = volume_of_mass(mj, cfg.rho_w), vk = volume_of_mass(mk, cfg.rho_w);
  const double rj = sj.r, rk = sk.r, uj = sj.u, uk = sk.u;
    switch (cfg.ec) {
      case SDM_EC_CONST: ec = cfg.ec_param[0]; break;
      case SDM_EC_BERRY1967: {
        const double e = linear_collection_efficiency(cfg.berry_params, rj, rk, cfg.berry_unit);
        ec = signed_sq(e);
        break;
      }
      case SDM_EC_LOWLIST1982: {  // coalescence_efficiencies/lowlist1982.py:37-103 (mass-based)
        const double PI = 3.14159265358979323846;
        const double ds = (rj < rk ? rj : rk) * 2, dl = (rj > rk ? rj : rk) * 2;
        double Sc = signed_pow(mj + mk, 2.0 / 3.0);
        Sc *= cfg.ec_param[1];  // PI * sgm_w * (6/PI)**(2/3), one host-side constant
        double St = ds * ds;
        St += dl * dl;
        St *= PI * cfg.sgm_w;
        const double dS = St - Sc;
        const double tmp = mj + mk;
        double tmp2 = fabs(uj - uk);
        tmp2 = tmp2 * tmp2;
        double CKE = mj * mk;
        if (tmp != 0.0) CKE /= tmp;
        CKE *= tmp2;
        CKE *= cfg.rho_w / 2;
        const double Et = CKE + dS;
        double e = signed_sq(Et);
        e *= -1.0 * 2.61e6 * cfg.sgm_w;
        e /= Sc;
        double o = ds / dl;
        o += 1.0;
        o = signed_pow(o, -2.0);
        o *= 0.778;
        o *= sdm_exp(e);
        ec = dl < 0.4e-3 ? 1.0 : o;
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
        ec = sdm_exp(We);
      }
    }
    switch (cfg.frag) {
      case SDM_FRAG_ALWAYS_N: fm = (mj + mk) / cfg.frag_param[0]; break;
      case SDM_FRAG_EXPONENTIAL: {
        const double a = 1 - u_b;
        double fv = -cfg.frag_param[0] * sdm_log(a > 1e-5 ? a : 1e-5), nf;
        fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, vj + vk);
        fm = cfg.rho_w * fv;
        break;
      }
      case SDM_FRAG_GAUSSIAN: {  // fragmentation_methods.py:477-485
        double fv = cfg.frag_param[0] + cfg.frag_param[1] *
                    erfinv_approx(u_b, cfg.straub_consts[3], cfg.straub_consts[4]), nf;
        fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, vj + vk);
        fm = cfg.rho_w * fv;
        break;
      }
      case SDM_FRAG_FEINGOLD1988: {  // :487-499, physics/fragmentation_function/feingold1988.py
        const double a = 1 - u_b * cfg.frag_param[0] / (vj + vk);
        double fv = -cfg.frag_param[0] * sdm_log(a > cfg.frag_param[1] ? a : cfg.frag_param[1]), nf;
        fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, vj + vk);
        fm = cfg.rho_w * fv;
        break;
      }
      case SDM_FRAG_SLAMS: {  // :95-134
        double p = 0.0, nf = 1;
        for (int k = 0; k < 22; ++k) {
          p += 0.91 * sdm_pow((double)(k + 2), -1.56);
          if (u_b < p) { nf = k + 2; break; }
        }
        double fv = (vj + vk) / nf;
        fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, vj + vk);
        fm = cfg.rho_w * fv;
        break;
      }
      case SDM_FRAG_CONSTANT_MASS: fm = cfg.frag_param[0]; break;  // constant_mass.py:11-14
      case SDM_FRAG_LOWLIST1982: {  // breakup_fragmentations/lowlist82.py:37-103 (volume-based)
        const double PI = 3.14159265358979323846;
        const double x_plus_y = vj + vk;
        const double ds = (rj < rk ? rj : rk) * 2, dl = (rj > rk ? rj : rk) * 2;
        double dcoal = x_plus_y / (PI / 6);
        dcoal = signed_pow(dcoal, 1.0 / 3.0);
        double Sc = signed_pow(x_plus_y, 2.0 / 3.0);
        Sc *= cfg.frag_param[0];  // PI * sgm_w * (6/PI)**(2/3), one host-side constant
        double St = ds * ds;
        St += dl * dl;
        St *= PI * cfg.sgm_w;
        double tmp2 = fabs(uj - uk);
        tmp2 = tmp2 * tmp2;
        double CKE = vj * vk;
        if (x_plus_y != 0.0) CKE /= x_plus_y;
        CKE *= tmp2;
        CKE *= cfg.rho_w / 2;
        double We = CKE, W2 = CKE;
        if (Sc != 0.0) We /= Sc;
        if (St != 0.0) W2 /= St;
        const double K[4] = {cfg.straub_consts[0], cfg.straub_consts[5], cfg.straub_consts[3],
                             cfg.straub_consts[4]};
        double rand = u_b, Rf = 0.0, Rs = 0.0, Rd = 0.0, nf;
        double fv = ll82_fragment_volume(CKE, We, W2, St, ds, dl, dcoal, &rand, &Rf, &Rs, &Rd,
                                         1e-8, K);
        fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, x_plus_y);
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
        double fv = straub_fragment_volume(CW, gam, ds, v_max, u_b, cfg.straub_consts, T), nf;
        fragmentation_limiters(nf, fv, cfg.frag_vmin, cfg.frag_nfmax, x_plus_y);
        fm = cfg.rho_w * fv;
      }
    }
}

struct PairInfo {
  bool have;
  uint8_t off;
  int64_t j, k, nj, nk, cid_j;
  double prob, dt_optimal;
};

// collision probability of one (multiplicity-sorted) pair in slot d: kernel, max multiplicity,
// normalisation (collision.py:249-254; cell of RAW super-droplet #d: reference quirk)
// `norm`: the slot's normalisation factor when the caller has it already (the per-cell kernel
// asks for it ahead of its LDS phases: three dependent look-ups off the critical path), else NULL
template <int KERNEL>
__device__ __forceinline__ double pair_prob_value(const sdm_step_cfg &cfg, const FusedArgs &A,
                                                  int64_t d, const SD &sj, const SD &sk,
                                                  const double *norm = nullptr) {
  const double vj = volume_of_mass(sj.m, cfg.rho_w), vk = volume_of_mass(sk.m, cfg.rho_w);
  double K;
  if (KERNEL == SDM_KERNEL_GOLOVIN) {
    K = (vj + vk) * cfg.kernel_param[0];
  } else if (KERNEL == SDM_KERNEL_GEOMETRIC) {
    const double s = sj.r + sk.r;
    K = (s * s) * cfg.kernel_param[0];
    K *= fabs(sj.u - sk.u);
  } else if (KERNEL == SDM_KERNEL_PARAMETERIZED) {  // parameterized.py:19-30, operation by operation
    const double e = linear_collection_efficiency(cfg.kernel_berry_params, sj.r, sk.r,
                                                  cfg.kernel_berry_unit);
    K = signed_sq(e);
    K *= 3.14159265358979323846;
    const double r_max = sj.r > sk.r ? sj.r : sk.r;
    K *= r_max * r_max;
    K *= fabs(sj.u - sk.u);
  } else if (KERNEL == SDM_KERNEL_SIMPLE_GEOMETRIC) {  // simple_geometric.py:21-27, area.py:14-17
    const double pi_4_3 = 3.14159265358979323846 * 4 / 3;
    const double aj = signed_pow(vj * (1 / pi_4_3), 2.0 / 3.0) * (pi_4_3 * 3);
    const double ak = signed_pow(vk * (1 / pi_4_3), 2.0 / 3.0) * (pi_4_3 * 3);
    const double s = sj.r + sk.r;
    K = cfg.kernel_param[0];
    K *= s * s;
    K *= fabs(aj - ak);
  } else if (KERNEL == SDM_KERNEL_LINEAR) {
    K = (vj + vk) * cfg.kernel_param[1];
    K += cfg.kernel_param[0];
  } else {
    K = cfg.kernel_param[0];
  }
  double prob = (double)sj.n;
  prob *= K;
  prob *= norm ? *norm
               : norm_factor_of(cfg, A.cell_start,
                                cfg.n_cell == 1 ? 0 : A.cell_idx[A.cell_id_raw[d]]);
  return prob;
}

// pairing (find_pairs + sort_within_pair), kernel, probability, [Ec, fragment mass], [optimal dt]
// for pair slot d; `u_b` = the slot's draw of the breakup streams
template <int KERNEL, bool BREAKUP>
__device__ __forceinline__ PairInfo pair_prob_body(const sdm_step_cfg &cfg, const FusedArgs &A,
                                                   int64_t d, int64_t W, double u_b) {
  PairInfo R;
  R.have = false; R.off = 2; R.j = R.k = R.nj = R.nk = R.cid_j = 0;
  R.prob = 0.0; R.dt_optimal = INFINITY;
  int64_t i = 0;
  int64_t tj = 0, tk = 0;
  // find_pairs (pair_methods.py:34-55) for positions 2d and 2d+1
  if (cfg.n_cell == 1) {
    if (2 * d + 1 < W) { R.have = true; i = 2 * d; }
    if (A.rec) {  // permutation resolved here (shuffle_local of the single cell [0, W))
      if (R.have) {
        walk_ids2(A.rec, A.rec_fmt, A.ovf_head, A.ovf_next, (int32_t)(2 * d),
                  (int32_t)(2 * d + 1), 0, tj, tk);
      } else {
        for (int o = 0; o < 2; ++o) {  // unpaired last position / dead tail
          const int64_t p = 2 * d + o;
          if (p < W) A.idx[p] = walk_id(A.rec, A.rec_fmt, A.ovf_head, A.ovf_next, (int32_t)p, 0);
          else if (p < cfg.n_sd) A.idx[p] = A.idx_prev[p];
        }
      }
    }
  } else {
#pragma unroll
    for (int o = 0; o < 2 && !R.have; ++o) {
      const int64_t p = 2 * d + o;
      if (p < W - 1) {
        const int64_t ca = A.cell_id[A.idx[p]], cb = A.cell_id[A.idx[p + 1]];
        const int64_t dd = p - A.cell_start[A.cell_idx[ca]];
        // (sharded runs on these generic kernels - cells beyond the per-cell kernels' capacity,
        // the global croupier: the pairs of another process's cell are that process's to collide)
        if (ca == cb && (dd & 1) == 0 && (!A.cell_owned || A.cell_owned[ca])) {
          R.have = true;
          i = p;
        }
      }
    }
  }
  if (!R.have) return R;
  R.off = (uint8_t)(i - 2 * d);
  const bool traced = A.rec != nullptr;
  int64_t j = traced ? tj : A.idx[i], k = traced ? tk : A.idx[i + 1];
  const bool need_r = KERNEL == SDM_KERNEL_GEOMETRIC || KERNEL == SDM_KERNEL_PARAMETERIZED ||
                      KERNEL == SDM_KERNEL_SIMPLE_GEOMETRIC;
  SD sj = sd_load(cfg, A, j, need_r), sk = sd_load(cfg, A, k, need_r);
  int64_t nj = sj.n, nk = sk.n;
  // sort_within_pair_by_attr (pair_methods.py:126-140)
  const bool swap = nj < nk;
  if (swap) {
    const int64_t t = j; j = k; k = t;
    const int64_t tn = nj; nj = nk; nk = tn;
    const SD ts = sj; sj = sk; sk = ts;
  }
  if (swap || traced) {
    A.idx[i] = j;
    A.idx[i + 1] = k;
  }
  R.j = j; R.k = k; R.nj = nj; R.nk = nk;
  R.cid_j = cfg.n_cell == 1 ? 0 : A.cell_id[j];
  const double prob = pair_prob_value<KERNEL>(cfg, A, d, sj, sk);
  R.prob = prob;
  if (cfg.adaptive && prob != 0) {
    // collisions_methods.py:359-368
    const int64_t prop = nj / nk;
    double t = cfg.dt * (double)prop / prob;
    R.dt_optimal = cfg.dt_min > t ? cfg.dt_min : t;
  }
  return R;
}

// compute_gamma + collision_coalescence[_breakup] for slot d (all lanes of the wave must call).
// p = the (already dt-scaled) probability; u = the slot's `rand`; j/k valid if `known`.
// one colliding pair: bounce / coalescence / breakup decision and the update
// (collisions_methods.py:247-311).  Wave-collective (counter_add): every lane calls it.
// pos: position of j in the permutation the step leaves behind (k sits at pos + 1)
struct Collided { int64_t j, k, cid, pos; double g, u_b; };

// returns which member ended with zero multiplicity (bit 0: the one passed as j, bit 1: as k)
template <bool BREAKUP>
__device__ __forceinline__ int resolve_collision(const sdm_step_cfg &cfg, const FusedArgs &A,
                                                 bool collide, int64_t j, int64_t k,
                                                 int64_t cid, double g, double u_b) {
  const int64_t j_in = j, k_in = k;
  const int64_t nk = collide ? A.multiplicity[k] : 0;
  bool coal = collide;
  // added to the counters once, at the end
  int64_t n_breakup = 0, n_breakup_deficit = 0, n_overflow = 0;
  if (BREAKUP && collide) {
    const double eb = cfg.eb_const;
    double ec, fm;
    {
      const bool need_ru = cfg.ec != SDM_EC_CONST || cfg.frag == SDM_FRAG_STRAUB2010 ||
                          cfg.frag == SDM_FRAG_LOWLIST1982;
      const SD sj = sd_load(cfg, A, j, need_ru), sk = sd_load(cfg, A, k, need_ru);
      breakup_params(cfg, A, sj, sk, u_b, ec, fm);
    }
    if (u_b - (ec + (1 - ec) * eb) > 0) {
      collide = false;  // bounce
      coal = false;
    } else if (!(u_b - ec < 0)) {
      coal = false;
      bool ovf = false;
      const double *mass = A.attributes + (int64_t)cfg.mass_attr * cfg.n_sd;
      // break_up / break_up_while (collisions_methods.py:135-243); counters through atomics
      double gamma_deficit = g;
      if (!cfg.handle_all_breakups) {
        double take_from_j, new_mult_k;
        int64_t gamma_j_k;
        compute_transfer_multiplicities(g, A.multiplicity[j], nk, mass[j], mass[k], fm,
                                        cfg.max_multiplicity, take_from_j, new_mult_k,
                                        gamma_j_k, ovf);
        gamma_deficit = g - (double)gamma_j_k;
        n_breakup += gamma_j_k * nk;
        if (gamma_deficit != 0) n_breakup_deficit += (int64_t)(gamma_deficit * (double)nk);
        apply_breakup_transfer(j, k, take_from_j, new_mult_k, A.multiplicity, A.attributes,
                               cfg.n_attr, cfg.n_sd);
      } else {
        while (gamma_deficit > 0) {
          double take_from_j, new_mult_k, gamma_j_k;
          const int64_t mj = A.multiplicity[j], mk = A.multiplicity[k];
          if (mk == mj) {
            take_from_j = (double)mj;
            new_mult_k = (mass[j] + mass[k]) / fm * (double)mk;
            if (new_mult_k > (double)cfg.max_multiplicity) {
              n_breakup_deficit += (int64_t)(gamma_deficit * (double)mk);
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
            // safety deviation: when not one breakup fits (the multiplicity limit or the donor's
            // count refuses the first already) the reference's loop never ends
            // (collisions_methods.py:192-236: gamma_deficit -= 0) - a hung CPU thread there, a
            // hung wavefront here; the rest goes to the deficit, as after any refused breakup
            if (g_int == 0) break;
          }
          const int64_t add = (int64_t)(gamma_j_k * (double)A.multiplicity[k]);
          n_breakup += add;
          gamma_deficit -= gamma_j_k;
          apply_breakup_transfer(j, k, take_from_j, new_mult_k, A.multiplicity, A.attributes,
                                 cfg.n_attr, cfg.n_sd);
        }
        const int64_t add = (int64_t)(gamma_deficit * (double)A.multiplicity[k]);
        n_breakup_deficit += add;
      }
      n_overflow = ovf ? 1 : 0;
    }
  }
  if (BREAKUP) {
    counter_add(A, CNT_BREAKUP, cid, n_breakup, true);
    counter_add(A, CNT_BREAKUP_DEFICIT, cid, n_breakup_deficit, true);
    counter_add(A, CNT_OVERFLOW, 0, n_overflow, true);
  }
  counter_add(A, CNT_COALESCENCE, cid, (int64_t)(g * (double)nk), coal);
  if (coal) coalesce_pair(j, k, g, A.multiplicity, A.attributes, cfg.n_attr, cfg.n_sd);
  int died = 0;
  if (collide) {
    const int64_t n_j = A.multiplicity[j_in], n_k = A.multiplicity[k_in];
    died = (n_j == 0 ? 1 : 0) | (n_k == 0 ? 2 : 0);
    if (died) A.ctl[CTL_HEALTHY] = 0;
    sd_refresh(cfg, A, j_in);
    sd_refresh(cfg, A, k_in);
  }
  return died;
}

// coalescence of a pair whose members' state is already in registers (one extensive attribute):
// same arithmetic as coalesce_pair (physics.h, collisions_methods.py:44-59) without re-reading
// anything; SoA columns and mirror records are written once.  Returns the died mask.
__device__ __forceinline__ int coalesce_known(const sdm_step_cfg &cfg, const FusedArgs &A,
                                              int64_t j, int64_t k, double g, SD sj, SD sk) {
  double *mass = A.attributes + (int64_t)cfg.mass_attr * cfg.n_sd;
  const double new_n = (double)sj.n - g * (double)sk.n;
  if (new_n > 0) {
    sj.n = (int64_t)new_n;
    sk.m += g * sj.m;
    A.multiplicity[j] = sj.n;
    mass[k] = sk.m;
    if (A.nm_wide) derive_ru(cfg, A, sk);
  } else {
    const int64_t half = sk.n / 2;
    sj.n = half;
    sk.n = sk.n - half;
    const double v = g * sj.m + sk.m;
    sj.m = sk.m = v;
    A.multiplicity[j] = sj.n;
    A.multiplicity[k] = sk.n;
    mass[j] = v;
    mass[k] = v;
    if (A.nm_wide) {
      derive_ru(cfg, A, sk);
      sj.r = sk.r;
      sj.u = sk.u;
    }
  }
  if (A.nm) {
    if (A.nm_wide) {
      ((double4 *)A.nm)[j] = make_double4(__longlong_as_double(sj.n), sj.m, sj.r, sj.u);
      ((double4 *)A.nm)[k] = make_double4(__longlong_as_double(sk.n), sk.m, sk.r, sk.u);
    } else {
      ((double2 *)A.nm)[j] = make_double2(__longlong_as_double(sj.n), sj.m);
      ((double2 *)A.nm)[k] = make_double2(__longlong_as_double(sk.n), sk.m);
    }
  }
  const int died = (sj.n == 0 ? 1 : 0) | (sk.n == 0 ? 2 : 0);
  if (died) A.ctl[CTL_HEALTHY] = 0;
  return died;
}

// the compaction of the fused route looks at the permutation only (index.hip, FLAG_ONLY): a
// super-droplet whose multiplicity reached zero is flagged where it sits, the way the reference
// flags precipitated ones (displacement_methods.py:157-158)
// sharded runs: `died` (mask of a pair) counted for the exchange - per process, and per segment
__device__ __forceinline__ void note_deaths(const FusedArgs &A, int64_t segment, int died) {
  const int k = (died & 1) + (died >> 1);
  if (A.n_dead) atomicAdd(A.n_dead, (unsigned long long)k);
  if (A.seg_deaths) atomicAdd(&A.seg_deaths[segment], -(double)k);
}

// `A` (may be NULL): where a sharded run counts its dead
__device__ __forceinline__ void flag_dead(const sdm_step_cfg &cfg, int64_t *__restrict__ idx,
                                          int64_t pos, int died, const FusedArgs *A = nullptr) {
  if (died & 1) idx[pos] = cfg.n_sd;
  if (died & 2) idx[pos + 1] = cfg.n_sd;
  if (A && died && A->dead_count) {
    const int k = (died & 1) + (died >> 1);
    const unsigned long long at = atomicAdd(A->dead_count, (unsigned long long)k);
    if (at + k <= SDM_DEAD_LIST_CAP) {
      if (died & 1) A->dead_pos[at] = pos;
      if (died & 2) A->dead_pos[at + (died & 1)] = pos + 1;
    }
  }
  if (A && died && (A->n_dead || A->seg_deaths))
    note_deaths(*A, A->seg_deaths ? find_cell(A->cell_start, cfg.n_cell, pos) : 0, died);
}


// `pos`: position of j in A.idx after this step's permutation (listed with the pair for
// k_resolve_dense); `flag_here`: flag dead members in A.idx right away (false: the caller holds the
// permutation elsewhere and applies the returned mask itself)
template <bool BREAKUP>
__device__ __forceinline__ int pair_update_body(const sdm_step_cfg &cfg, const FusedArgs &A,
                                                int64_t d, bool in_range, double p, double u,
                                                double u_b, bool known, int64_t off,
                                                int64_t j, int64_t k, int64_t pos,
                                                bool flag_here, const SD *sj = nullptr,
                                                const SD *sk = nullptr) {
  bool collide = false;
  int64_t cid = 0, nk = 0, gi = 0, gc = 0;
  double g = 0;
  if (in_range) {
    g = ceil(p - u);  // collisions_methods.py:560
    // pair_indices' skip (gamma == 0) also covers "no pair": then prob == 0, gamma = -0.0
    if (g != 0 && off < 2) {
      collide = true;
      if (!known) {
        j = A.idx[2 * d + off];
        k = A.idx[2 * d + 1 + off];
      }
      nk = sk ? sk->n : A.multiplicity[k];
      const int64_t prop = (sj ? sj->n : A.multiplicity[j]) / nk;
      gi = (int64_t)g;
      gc = gi < prop ? gi : prop;
      cid = cfg.n_cell == 1 ? 0 : A.cell_id[j];
      g = (double)gc;
    }
  }
  counter_add(A, CNT_COLLISION, cid, gc * nk, collide);
  counter_add(A, CNT_COLLISION_DEFICIT, cid, (gi - gc) * nk, collide);
  collide = collide && g != 0;
  if (BREAKUP) {
    // breakup: the (rare, transcendental-heavy) resolution runs densely packed in
    // k_resolve_dense; here the colliding pairs are only listed
    const unsigned long long m = __ballot(collide);
    if (m != 0) {
      const int lane = lane_id(), leader = __ffsll((long long)m) - 1;
      unsigned long long base = 0;
      const int64_t l = blockIdx.x % (unsigned)A.list_nl;
      if (lane == leader) base = atomicAdd(A.list_count + l * SDM_CNT_STRIDE,
                                           (unsigned long long)__popcll(m));
      base = __shfl((long long)base, leader, 64);
      if (collide) {
        Collided c;
        c.j = j; c.k = k; c.cid = cid; c.pos = pos; c.g = g; c.u_b = u_b;
        A.list[l * A.list_cap + base + __popcll(m & ((1ull << lane) - 1))] = c;
      }
    }
    return 0;
  }
  int died;
  if (sj && cfg.n_attr == 1) {  // members' state carried in registers by the caller
    counter_add(A, CNT_COALESCENCE, cid, (int64_t)(g * (double)nk), collide);
    died = collide ? coalesce_known(cfg, A, j, k, g, *sj, *sk) : 0;
  } else {
    died = resolve_collision<false>(cfg, A, collide, j, k, cid, g, u_b);
  }
  if (died && flag_here) flag_dead(cfg, A.idx, pos, died, &A);
  return died;
}

// ---- non-adaptive: everything about a pair in one kernel -------------------------------------
template <int KERNEL, bool BREAKUP>
__global__ void __launch_bounds__(SDM_BLOCK) k_pair_all(sdm_step_cfg cfg, FusedArgs A) {
  __shared__ u128 lds[2];
  const int64_t W = A.ctl[CTL_WORK];
  const int64_t d = TID();
  if (d == 0) A.ctl[CTL_PAIRS] += W / 2;  // lets a caller count pairs without reading back per step
  const double u = stream_draw(A.s_rand, A.rng_inc, A.rng_tab, &lds[0],
                               A.dev_off ? A.dev_off[0] + A.rand_extra : 0, A.rng_aff);
  const double u_b = BREAKUP ? stream_draw(A.s_rand_b, A.rng_inc, A.rng_tab, &lds[1],
                                           A.dev_off ? A.dev_off[1] : 0, A.rng_aff)
                             : 0.0;
  PairInfo R;
  R.have = false; R.off = 2; R.prob = 0; R.j = R.k = 0;
  if (d < (cfg.n_sd + 1) / 2) R = pair_prob_body<KERNEL, BREAKUP>(cfg, A, d, W, u_b);
  double p = R.prob;
  if (p != 0) p /= (double)cfg.substeps;  // collision.py:279
  pair_update_body<BREAKUP>(cfg, A, d, d < W / 2, p, u, u_b, true, R.off, R.j, R.k,
                            2 * d + R.off, true);
}

// The same with the tile sort of the NEXT sub-step's shuffle build on the side: the first
// X.n_tiles workgroups sort (index.hip: k_bin_sort's body - generator arithmetic and LDS, which a
// kernel that waits on random memory accesses has to spare), the others are the pair kernel in
// workgroups of BIN_THREADS.  The sort is for the length this sub-step begins with; if a
// super-droplet dies in it, the next build redoes it after the compaction (k_bin_build2).  Saves
// that sub-step a 13-us kernel with its boundary (coalescence only: the pair lists of the
// breakup route are sized for SDM_BLOCK-thread workgroups).
struct SortAhead {
  int2 *events;
  int32_t *toff, *jarr, *loc;
  int n_bins, n_tiles;
  const int64_t *p_length;
  u128 s_off;  // generator state at the next sub-step's first u01
};

// draw number `index` of the stream starting at s_base (ready jumps: common.h, pcg_aff)
__device__ __forceinline__ double draw_at(u128 s_base, u128 inc, const u128 *__restrict__ aff,
                                          int64_t index) {
  u128 state = pcg_apply(pcg_apply(s_base, aff, PCG_AFF_SMALL + index / PCG_AFF_STRIDE), aff,
                         index % PCG_AFF_STRIDE);
  state = state * pcg_mult() + inc;
  return pcg_output(state);
}

template <int KERNEL>
__global__ void __launch_bounds__(BIN_THREADS)
k_pair_all_sort(sdm_step_cfg cfg, FusedArgs A, SortAhead X) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < X.n_tiles) {
    const int64_t length = *X.p_length;
    bin_sort_body<true>(smem, X.events, X.toff, X.jarr, X.loc, X.n_bins, nullptr, nullptr, 1,
                        length, length, X.s_off, A.rng_inc, A.rng_tab, nullptr, A.rng_aff);
    return;
  }
  const int64_t W = A.ctl[CTL_WORK];
  const int64_t d = (int64_t)(blockIdx.x - X.n_tiles) * BIN_THREADS + threadIdx.x;
  if (d == 0) A.ctl[CTL_PAIRS] += W / 2;
  const double u = draw_at(A.s_rand, A.rng_inc, A.rng_aff, d);
  PairInfo R;
  R.have = false; R.off = 2; R.prob = 0; R.j = R.k = 0;
  if (d < (cfg.n_sd + 1) / 2) R = pair_prob_body<KERNEL, false>(cfg, A, d, W, 0.0);
  double p = R.prob;
  if (p != 0) p /= (double)cfg.substeps;  // collision.py:279
  pair_update_body<false>(cfg, A, d, d < W / 2, p, u, 0.0, true, R.off, R.j, R.k, 2 * d + R.off,
                          true);
}

// ---- adaptive: probabilities first (per-cell min of the optimal dt is a global dependency) ---
template <int KERNEL, bool BREAKUP>
__global__ void __launch_bounds__(SDM_BLOCK) k_pair_prob(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t W = A.ctl[CTL_WORK];
  const int64_t d = TID();
  const double u_b = 0.0;  // breakup parameters are evaluated in k_pair_update
  PairInfo R;
  R.have = false; R.prob = 0; R.cid_j = 0; R.dt_optimal = INFINITY; R.off = 2;
  if (d < (cfg.n_sd + 1) / 2) R = pair_prob_body<KERNEL, BREAKUP>(cfg, A, d, W, u_b);
  if (d < cfg.n_sd / 2) {
    A.prob[d] = R.prob;
    A.pair_off[d] = R.off;
    if (cfg.n_cell > 1) A.pair_cid[d] = (int32_t)R.cid_j;
  }
  const bool active = R.have && R.prob != 0;
  if (cfg.n_cell == 1) {
    // one cell: no atomics on a single word; workgroup minimum -> block_min[blockIdx]
    __shared__ double wmin[SDM_BLOCK / SDM_WAVE];
    const double m = wave_min_f64(active ? R.dt_optimal : INFINITY);
    if (lane_id() == 0) wmin[threadIdx.x / SDM_WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      double b = wmin[0];
      for (int w = 1; w < SDM_BLOCK / SDM_WAVE; ++w) b = wmin[w] < b ? wmin[w] : b;
      A.block_min[blockIdx.x] = b;
    }
    return;
  }
  const unsigned long long am = __ballot(active);
  if (am != 0) {
    const int first = __ffsll((long long)am) - 1;
    const int64_t cid0 = __shfl((long long)R.cid_j, first, 64);
    if (__all(!active || R.cid_j == cid0)) {
      const double m = wave_min_f64(active ? R.dt_optimal : INFINITY);
      if (lane_id() == first) atomic_min_pos_f64(&A.cell_min[cid0], m);
    } else if (active) {
      atomic_min_pos_f64(&A.cell_min[R.cid_j], R.dt_optimal);
    }
  }
}

// ---- per-cell adaptive bookkeeping (collisions_methods.py:357-374) ---------------------------
// pre: 0 = dt_todo / dt_left were prepared by k_cells_pre; 1 = do its part here (one cell: no
// kernel in between needs them); 2 = likewise, and this is the first sub-step of the time step
// (dt_left[:] = dt, collision.py:180)
__global__ void __launch_bounds__(SDM_BLOCK) k_cells_adaptive(sdm_step_cfg cfg, FusedArgs A,
                                                               int pre) {
  if (cfg.n_cell == 1) {  // fold the per-workgroup partial minima (one workgroup launched)
    __shared__ double wmin[SDM_BLOCK / SDM_WAVE];
    double m = INFINITY;
    // (eight loads in flight per thread: 8192 partial minima at 2^22 super-droplets took 13 us
    // one dependent-looking load at a time)
    for (int b0 = threadIdx.x; b0 < A.n_block_min; b0 += 8 * SDM_BLOCK) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int b = b0 + k * SDM_BLOCK;
        v[k] = b < A.n_block_min ? A.block_min[b] : INFINITY;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) m = v[k] < m ? v[k] : m;
    }
    m = wave_min_f64(m);
    if (lane_id() == 0) wmin[threadIdx.x / SDM_WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < SDM_BLOCK / SDM_WAVE; ++w) m = wmin[w] < m ? wmin[w] : m;
      A.cell_min[0] = m;
    }
    __syncthreads();
  }
  const int64_t c = TID();
  if (c >= cfg.n_cell) return;
  if (A.ctl[CTL_WORK] == 0) return;
  const double m = ((volatile double *)A.cell_min)[c];
  const double l = pre == 2 ? cfg.dt : A.dt_left[c];
  double t = pre ? (cfg.dt_max < l ? cfg.dt_max : l) : A.dt_todo[c];  // Python min(l, dt_max)
  if (m < t) t = m;
  A.dt_todo[c] = t;
  const double s = A.stats_dt_min[c];
  const double s_new = m < s ? m : s;  // Python min(s, m): NaN-sticky
  A.stats_dt_min[c] = s_new;
  note_dt_min(A.ctl, s_new, cfg.dt_min);
  A.dt_left[c] = l - t;
  if (t > 0) A.stats_n_substep[c] += 1;
}

// ---- reset_cell_idx of a state that is grouped by cell (collision.py:190) ---------------------
// At the end of an adaptive time step cell_idx becomes the identity and the permutation has to be
// sorted by it again.  The super-droplets of each cell are contiguous at that point (the state is
// sorted by the cell_idx of the last counting sort), so the stable counting sort by cell id only
// moves whole segments: segment k (cell_start[k] .. cell_start[k+1]) belongs to the cell of its
// first member.  Sizes by cell id -> exclusive scan = new cell_start -> segment copies into the
// spare permutation buffer: streaming only (30 us at 2^22 super-droplets in 1024 cells, against
// 150 us for the counting sort with its gather of one cell id per super-droplet).
__global__ void __launch_bounds__(SDM_BLOCK)
k_reseg_sizes(const int64_t *__restrict__ idx, const int64_t *__restrict__ cell_id,
              const int64_t *__restrict__ cell_start, int64_t n_cell,
              int64_t *__restrict__ seg_size, int64_t *__restrict__ seg_src) {
  const int64_t k = TID();
  if (k >= n_cell) return;
  const int64_t a = cell_start[k], b = cell_start[k + 1];
  if (b > a) {
    const int64_t c = cell_id[idx[a]];
    seg_size[c] = b - a;
    seg_src[c] = a;
  }
}

// workgroup c < n_cell: segment of cell c to its new place; the others: dead tail [valid, n_sd)
__global__ void __launch_bounds__(SDM_BLOCK)
k_reseg_copy(int64_t *__restrict__ out, const int64_t *__restrict__ idx,
             const int64_t *__restrict__ seg_size, const int64_t *__restrict__ seg_src,
             const int64_t *__restrict__ cell_start_new, int64_t n_cell, int64_t n_sd,
             int64_t *ctl) {
  const int64_t b = blockIdx.x;
  if (b < n_cell) {
    const int64_t n = seg_size[b], from = seg_src[b], to = cell_start_new[b];
    for (int64_t t = threadIdx.x; t < n; t += SDM_BLOCK) out[to + t] = idx[from + t];
    if (b == 0 && threadIdx.x == 0) ctl[CTL_SORTED] = 1;
    return;
  }
  const int64_t n_tail = gridDim.x - n_cell;
  for (int64_t i = ctl[CTL_VALID] + (b - n_cell) * SDM_BLOCK + threadIdx.x; i < n_sd;
       i += n_tail * SDM_BLOCK)
    out[i] = idx[i];
}

// ---- multi-cell per-cell route: what ends one sub-step and opens the next, in ONE launch ---------
// k_cells_turn, one workgroup per cell id i.  Between two cell kernels an adaptive step needs: the
// per-cell bookkeeping of the sub-step just done (collisions_methods.py:357-374: dt_left, stats),
// adaptive_sdm_end (:313-328: the working length), the decision whether another sub-step runs, and
// for that one cell_idx.sort_by_key(dt_left) (collision.py:183) and the per-cell init (:355-356).
// Rounds 2-3 did this in two launches around a finish ticket (per-cell bookkeeping + an atomicMax
// in the compaction kernel's workgroups, the ranking in a kernel of its own).  Here EVERY workgroup
// recomputes what is global - all cells' new dt_left (n_cell values: four per thread at 1024
// cells), from which it takes the rank of its own cell, the end of the working range and the
// events - from inputs nobody writes in this launch, and writes its own cell's results only: no
// atomics, no ticket, no fence, one launch; workgroup 0 also writes the control words and
// publishes the control block of the sub-step that ended.  Hence the ping-pong: dt_left is read
// from `left_in` and written to `left_out`, the cell minima are read from `min_in` (filled by the
// previous cell kernel and - sharded runs - completed by the exchange) and this sub-step's are
// reset in `min_out`.
// min_*[c]: the minimum of the optimal sub-step over cell c's pairs (+inf: none / not this
// process's cell); min_*[n_cell + k]: minus the number of super-droplets that died in segment k
// (sharded runs; an all-reduce MIN over the processes completes both halves at once).
// gate[(turn) & 1]: whether the sub-step this launch opens runs (read by its cell kernel);
// gate[(turn - 1) & 1]: whether the previous one ran (then its bookkeeping is applied here).
struct TurnArgs {
  const double *left_in;
  double *left_out;
  const double *min_in;
  double *min_out;
  int64_t *cell_idx;
  int64_t *gate;
  int64_t turn;      // launch counter of the call (parity selects the gate word)
  int32_t first;     // no previous sub-step in this call: nothing to apply
  int32_t fresh;     // first sub-step of a time step: dt_left[:] = dt (collision.py:180)
  int32_t gated;     // the sub-step was launched ahead: it runs only if work is left, the state is
                     // sorted and nobody died (a death is dealt with by the host: compaction, re-sort)
  int32_t end_only;  // only end the previous sub-step (no launch ahead: timing mode)
  int32_t sharded;   // the death counts in min_in are meaningful
  int64_t *box;
  int64_t seq;       // publication number of the sub-step that ended (0: nothing to publish)
  // sharded runs: the permutation the cell kernel will read, and where "this segment's cell is
  // mine" goes (FusedArgs::seg_owned); NULL otherwise
  const int64_t *perm;
  uint8_t *seg_owned;
  int32_t *seg_cid;  // (any run: FusedArgs::seg_cid)
};

// TURN_GROUP cells per workgroup: what is read of all cells - their dt_left and minima - serves
// the ranks of the group's cells together.  One cell per workgroup is fastest at 1024 cells (7.9
// against 10.6 us with eight: the kernel is a chain of latencies there, and the eight-fold compare
// lengthens it), but the work grows with the SQUARE of the number of cells: 45 us at 75 x 75 =
// 5625 cells, more than that grid's cell kernel (40 us) - eight per workgroup above 2048 cells,
// sixteen above 8192 (measured at 5625: sixteen 29 us against eight 21 - fewer, fatter wavefronts)
template <int TURN_GROUP>
__global__ void __launch_bounds__(SDM_BLOCK) k_cells_turn(sdm_step_cfg cfg, FusedArgs A, TurnArgs T) {
  __shared__ int sm_rank[SDM_BLOCK / SDM_WAVE][TURN_GROUP];
  __shared__ int sm_top[SDM_BLOCK / SDM_WAVE], sm_flags[SDM_BLOCK / SDM_WAVE];
  const int64_t n = cfg.n_cell, i0 = (int64_t)blockIdx.x * TURN_GROUP;
  // every load the kernel needs is issued before anything depends on one: the kernel is a chain
  // of memory round trips (gate -> inputs -> cell_start[top] -> control words), 1.5-2 us each,
  // and nothing else; which loads are needed follows from the launch arguments alone
  const bool may_apply = !T.first;
  const bool read_in = !T.fresh;
  // cells per thread and pass: SDM_BLOCK * PER >= 1024 cells at once; more on large grids, where
  // the passes' loads are what the kernel waits for
  constexpr int PER = TURN_GROUP == 1 ? 4 : 8;
  const int64_t gate_w = may_apply ? T.gate[(T.turn - 1) & 1] : 0;
  const int64_t ctl_work = A.ctl[CTL_WORK], ctl_healthy = A.ctl[CTL_HEALTHY],
                ctl_sorted = A.ctl[CTL_SORTED];
  if (T.seg_cid) {  // (a chain of three or four loads of its own, beside everything else)
    const int g = (int)threadIdx.x - SDM_WAVE;
    if (g >= 0 && g < TURN_GROUP && i0 + g < n) {
      const int64_t lo = A.cell_start[i0 + g], hi = A.cell_start[i0 + g + 1];
      // (a sub-step launched ahead of the host's knowledge of a death finds FLAGGED entries - the
      // id n_sd - in the permutation: its gate will close, but this look-up runs regardless)
      const int64_t id = hi > lo && lo >= 0 && lo < cfg.n_sd ? T.perm[lo] : -1;
      const int64_t cid = id >= 0 && id < cfg.n_sd ? A.cell_id[id] : -1;
      T.seg_cid[i0 + g] = (int32_t)cid;
      if (T.seg_owned) T.seg_owned[i0 + g] = cid >= 0 ? A.cell_owned[cid] : 0;
    }
  }
  double own_l[TURN_GROUP], own_m[TURN_GROUP];
#pragma unroll
  for (int g = 0; g < TURN_GROUP; ++g) {
    const bool in = i0 + g < n;
    own_l[g] = in && read_in ? T.left_in[i0 + g] : cfg.dt;
    own_m[g] = in && read_in && may_apply ? T.min_in[i0 + g] : INFINITY;
  }
  const bool apply = may_apply && gate_w != 0;
  // new dt_left of a cell from its old one and its minimum (what the bookkeeping of the sub-step
  // just done leaves)
  auto left_of = [&](double l, double m, double *todo) -> double {
    if (T.fresh) return cfg.dt;
    if (!apply) return l;
    double t = cfg.dt_max < l ? cfg.dt_max : l;  // Python min(l, dt_max)
    if (m < t) t = m;
    if (todo) *todo = t;
    return l - t;
  };
  double left_i[TURN_GROUP];
#pragma unroll
  for (int g = 0; g < TURN_GROUP; ++g)
    left_i[g] = i0 + g < n ? left_of(own_l[g], own_m[g], nullptr) : 0.0;
  int rank[TURN_GROUP];
#pragma unroll
  for (int g = 0; g < TURN_GROUP; ++g) rank[g] = 0;
  int64_t end = 0;  // adaptive_sdm_end (collisions_methods.py:313-328): dt_left is scanned by
                    // POSITION there; cell_start is monotone, so the largest cell_start[c + 1]
                    // over the cells with time left is the one of the last such cell
  int flags = 0;    // 1 = somebody died, 2 = a cell minimum equals dt_min
  // The ranking is n_cell^2 comparisons (31.6 M at 75 x 75 cells): it is what the kernel costs on a
  // large grid, so a comparison has to be ONE instruction.  rank(i) = #{c: left_c < left_i} +
  // #{c < i: left_c == left_i}; for the cells BELOW the group's own range that is `<=`, for those
  // ABOVE it `<`, and only the group's own cells need the index - three loops over cells, each
  // with one uniform test (MODE 0 / 1 / 2), instead of two floating-point compares, a 64-bit
  // integer compare and the logic between them per pair (21 -> ~9 us at 5625 cells)
  auto scan = [&](int64_t c_begin, int64_t c_end, auto mode) {
    constexpr int MODE = decltype(mode)::value;
    for (int64_t c0 = c_begin; c0 < c_end; c0 += SDM_BLOCK * PER) {
      double l[PER], m[PER], d[PER];
      int64_t e[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int64_t c = c0 + k * SDM_BLOCK + threadIdx.x;
        const bool in = c < c_end;
        l[k] = in && read_in ? T.left_in[c] : cfg.dt;
        m[k] = in && read_in && may_apply ? T.min_in[c] : INFINITY;
        d[k] = in && read_in && may_apply && T.sharded ? T.min_in[n + c] : 0.0;
        // (many cells: the end of the working range is found as an INDEX and looked up once)
        e[k] = !in ? 0 : (TURN_GROUP == 1 ? A.cell_start[c + 1] : c + 1);
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int64_t c = c0 + k * SDM_BLOCK + threadIdx.x;
        if (c >= c_end) continue;
        const double lc = left_of(l[k], m[k], nullptr);
#pragma unroll
        for (int g = 0; g < TURN_GROUP; ++g) {
          if (MODE == 0) rank[g] += lc <= left_i[g];
          else if (MODE == 1) rank[g] += lc < left_i[g];
          else rank[g] += (lc < left_i[g]) || (lc == left_i[g] && c < i0 + g);
        }
        if (lc != 0 && e[k] > end) end = e[k];
        if (apply) {
          if (d[k] < 0) flags |= 1;
          if (m[k] == cfg.dt_min) flags |= 2;
        }
      }
    }
  };
  if (TURN_GROUP == 1) {  // (few cells: a chain of latencies - one pass, all loads at once)
    scan(0, n, std::integral_constant<int, 2>());
  } else {
    const int64_t own_end = i0 + TURN_GROUP < n ? i0 + TURN_GROUP : n;
    scan(0, i0, std::integral_constant<int, 0>());
    scan(i0, own_end, std::integral_constant<int, 2>());
    scan(own_end, n, std::integral_constant<int, 1>());
  }
#pragma unroll
  for (int g = 0; g < TURN_GROUP; ++g) rank[g] = wave_sum_i32(rank[g]);
  int top = (int)end;  // (positions fit 31 bits: n_sd < 2^31 is checked at the boundary)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t2 = __shfl_xor(top, o, 64);
    top = t2 > top ? t2 : top;
    flags |= __shfl_xor(flags, o, 64);
  }
  if (lane_id() == 0) {
#pragma unroll
    for (int g = 0; g < TURN_GROUP; ++g) sm_rank[threadIdx.x / SDM_WAVE][g] = rank[g];
    sm_top[threadIdx.x / SDM_WAVE] = top;
    sm_flags[threadIdx.x / SDM_WAVE] = flags;
  }
  __syncthreads();
  top = flags = 0;
  for (int w = 0; w < SDM_BLOCK / SDM_WAVE; ++w) {
    top = sm_top[w] > top ? sm_top[w] : top;
    flags |= sm_flags[w];
  }
  end = TURN_GROUP == 1 ? (int64_t)top : (top == 0 ? 0 : A.cell_start[top]);
  const int64_t work = apply ? end : ctl_work;
  const bool healthy = ctl_healthy != 0 && !(flags & 1);
  const bool run = !T.end_only && (!T.gated || (work != 0 && ctl_sorted != 0 && healthy));
  // the group's own results: thread g writes those of cell i0 + g
  const int g = threadIdx.x;
  if (g < TURN_GROUP && i0 + g < n) {
    const int64_t i = i0 + g;
    double t_i = 0.0;
    double l_g = own_l[0], m_i = own_m[0];  // (own_l[g] with a run-time g would spill)
#pragma unroll
    for (int k = 1; k < TURN_GROUP; ++k)
      if (k == g) { l_g = own_l[k]; m_i = own_m[k]; }
    const double mine = left_of(l_g, m_i, &t_i);
    if (!apply) m_i = INFINITY;
    int total = 0;
    for (int w = 0; w < SDM_BLOCK / SDM_WAVE; ++w) total += sm_rank[w][g];
    T.left_out[i] = mine;
    if (apply) {
      const double smin = A.stats_dt_min[i];
      A.stats_dt_min[i] = m_i < smin ? m_i : smin;  // Python min(s, m): NaN-sticky
      if (t_i > 0) A.stats_n_substep[i] += 1;
    }
    if (run) {
      T.cell_idx[n - 1 - total] = i;
      T.min_out[i] = INFINITY;
      if (T.sharded) T.min_out[n + i] = 0.0;
    }
  }
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  T.gate[T.turn & 1] = run ? 1 : 0;
  if (apply) A.ctl[CTL_WORK] = end;
  if (flags & 1) A.ctl[CTL_HEALTHY] = 0;
  // the reference's warning "adaptive time-step reached dt_min" (collision.py:276-277): the host
  // evaluates amin(stats_dt_min) == dt_min when this bit is set.  A minimum equal to dt_min is
  // what can make it so (the optimal sub-step is clamped to dt_min from below)
  if (flags & 2) A.ctl[7] |= SDM_CTL7_DT_MIN;
  if (T.seq) publish_ctl(A.ctl, T.box, T.seq, work);
}

template <bool BREAKUP>
__global__ void __launch_bounds__(SDM_BLOCK) k_pair_update(sdm_step_cfg cfg, FusedArgs A) {
  __shared__ u128 lds[2];
  const int64_t W = A.ctl[CTL_WORK];
  const int64_t d = TID();
  const double u = stream_draw(A.s_rand, A.rng_inc, A.rng_tab, &lds[0], 0, A.rng_aff);
  const double u_b = BREAKUP ? stream_draw(A.s_rand_b, A.rng_inc, A.rng_tab, &lds[1], 0,
                                           A.rng_aff)
                             : 0.0;
  const bool in_range = d < W / 2;
  // One adaptive cell: the per-cell bookkeeping of collisions_methods.py:357-374 needs the minimum
  // over ALL pairs - a kernel boundary after k_pair_prob - but not a kernel: every workgroup folds
  // the partial minima for itself (16 KB from L2 at 2^20 super-droplets, requested before the
  // stored probability is), workgroup 0 writes the cell's words.  Saves the k_cells_adaptive
  // launch (4.9 us + a boundary per sub-step)
  double todo = 0;
  if (A.fold_pre) {
    __shared__ double wmin[SDM_BLOCK / SDM_WAVE];
    __shared__ double s_todo;
    double m = INFINITY;
    for (int b0 = threadIdx.x; b0 < A.n_block_min; b0 += 8 * SDM_BLOCK) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int b = b0 + k * SDM_BLOCK;
        v[k] = b < A.n_block_min ? A.block_min[b] : INFINITY;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) m = v[k] < m ? v[k] : m;
    }
    m = wave_min_f64(m);
    if (lane_id() == 0) wmin[threadIdx.x / SDM_WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < SDM_BLOCK / SDM_WAVE; ++w) m = wmin[w] < m ? wmin[w] : m;
      const double l = A.fold_pre == 2 ? cfg.dt : A.dt_left_pub[0];
      double t = cfg.dt_max < l ? cfg.dt_max : l;  // Python min(l, dt_max)
      if (m < t) t = m;
      s_todo = t;
      if (blockIdx.x == 0) {
        A.cell_min[0] = m;
        if (W != 0) {
          A.dt_todo[0] = t;
          const double s = A.stats_dt_min[0];
          const double s_new = m < s ? m : s;  // Python min(s, m): NaN-sticky
          A.stats_dt_min[0] = s_new;
          note_dt_min(A.ctl, s_new, cfg.dt_min);
          A.dt_left[0] = l - t;
          if (t > 0) A.stats_n_substep[0] += 1;
        }
      }
    }
    __syncthreads();
    todo = s_todo;
  }
  double p = 0;
  int64_t off = 2;
  if (in_range) {
    p = A.prob[d];
    off = A.pair_off[d];
    if (p != 0) {
      const int64_t cid = cfg.n_cell > 1 ? A.pair_cid[d] : 0;
      p *= (A.fold_pre ? todo : A.dt_todo[cid]) / cfg.dt;  // collisions_methods.py:369-372
    }
  }
  pair_update_body<BREAKUP>(cfg, A, d, in_range, p, u, u_b, false, off, 0, 0, 2 * d + off, true);
}

// ---- multi-cell fast path: one workgroup per cell, the whole sub-step of the cell in LDS ---------
// For cells of at most CELL_CAP super-droplets: shuffle_local's events, hit lists and the backward
// walks live in LDS (no binning passes, no global scatter); pairing needs no per-droplet cell-id
// gathers (a sorted segment holds one cell: same key <=> same cell id); the per-cell minimum of the
// optimal sub-step is a workgroup reduction, so probability, gamma and update are one kernel also
// in adaptive mode.  Global random traffic left: one mirror record per droplet.
#ifndef CELL_CAP
#define CELL_CAP 6144
#endif
#ifndef CELL_THREADS
#define CELL_THREADS 1024
#endif
#define CELL_MAXPOS (CELL_CAP / CELL_THREADS)
#define CELL_MAXPAIR (CELL_CAP / 2 / CELL_THREADS)
#define CELL_LDS_BYTES (5 * CELL_CAP * 4 + 2 * CELL_CAP * 2)

struct CellArgs {
  const int64_t *idx_in;
  int64_t *idx_out;
  u128 s_u01;  // PCG64 state at draw 0 of the sub-step's u01 window
  int n_tail_blocks;
  const int64_t *gate;  // NULL, or a word written by k_cells_turn: 0 = this sub-step does not run
  // sharded mode: segments of cells this process does not own are copied through unchanged (their
  // content is only required to hold the cell's members) - needed when idx_out is not known to
  // hold them already
  int copy_others;
};

#ifdef CELL_PROFILE
#define CELL_MARK(k) do { __syncthreads(); if (blockIdx.x == 7 && threadIdx.x == 0) cell_t[(k) + 1] = wall_clock64(); } while (0)
#else
#define CELL_MARK(k)
#endif

template <int KERNEL, bool BREAKUP>
__global__ void __launch_bounds__(CELL_THREADS)
k_cell_step(sdm_step_cfg cfg, FusedArgs A, CellArgs X) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef CELL_PROFILE
  __shared__ long long cell_t[9];
  if (blockIdx.x == 7 && threadIdx.x == 0) cell_t[0] = wall_clock64();
#endif
  int32_t *s0 = (int32_t *)smem, *s1 = s0 + CELL_CAP, *head = s1 + CELL_CAP;
  int32_t *val = head + CELL_CAP, *out = val + CELL_CAP;
  int16_t *jown = (int16_t *)(out + CELL_CAP), *next = jown + CELL_CAP;
  __shared__ double red[CELL_THREADS / SDM_WAVE];
  __shared__ int64_t s_cid, s_base;
  __shared__ u128 s_rng[3];  // generator states at the cell's first position / first pair slot
  const int64_t C = cfg.n_cell, N = cfg.n_sd;
  const int tid = threadIdx.x;
  if (X.gate && X.gate[0] == 0) return;
  if ((int64_t)blockIdx.x >= C) {  // dead tail [cell_start[C], N) is carried over unchanged
    const int64_t from = A.cell_start[C];
    for (int64_t i = from + ((int64_t)blockIdx.x - C) * CELL_THREADS + tid; i < N;
         i += (int64_t)X.n_tail_blocks * CELL_THREADS)
      X.idx_out[i] = X.idx_in[i];
    return;
  }
  const int64_t lo = A.cell_start[blockIdx.x], hi = A.cell_start[blockIdx.x + 1];
  const int n = (int)(hi - lo);
  if (n == 0) return;
  if (A.cell_owned && !A.cell_owned[A.cell_id[X.idx_in[lo]]]) {  // another process's cell
    if (X.copy_others)
      for (int li = tid; li < n; li += CELL_THREADS) X.idx_out[lo + li] = X.idx_in[lo + li];
    return;
  }
  if (n > CELL_CAP) {  // never taken: the host enables this path only below the cap
    if (tid == 0) A.ctl[7] = 1;
    for (int li = tid; li < n; li += CELL_THREADS) X.idx_out[lo + li] = X.idx_in[lo + li];
    return;
  }
  CELL_MARK(0);
  const int64_t W = A.ctl[CTL_WORK];
  for (int li = tid; li < n; li += CELL_THREADS) {
    val[li] = (int32_t)X.idx_in[lo + li];
    s0[li] = -1; s1[li] = -1; head[li] = -1;
  }
  __syncthreads();
  if (tid == 0) {
    s_cid = A.cell_id[val[0]];
    s_base = A.cell_start[A.cell_idx[s_cid]];
  }
  // long jump-aheads once per workgroup (three wavefronts, one each); threads then only jump by
  // their small offset inside the cell
  if (tid == 64) s_rng[0] = pcg_jump_fast(X.s_u01, A.rng_tab, A.rng_aff, (uint64_t)lo);
  if (tid == 128) s_rng[1] = pcg_jump_fast(A.s_rand, A.rng_tab, A.rng_aff, (uint64_t)(lo >> 1));
  if (BREAKUP && tid == 192) s_rng[2] = pcg_jump_fast(A.s_rand_b, A.rng_tab, A.rng_aff, (uint64_t)(lo >> 1));
  __syncthreads();
  CELL_MARK(1);
  // shuffle_local events of this cell (index_methods.py:35-41): consecutive positions per thread
  {
    const int chunk = (n + CELL_THREADS - 1) / CELL_THREADS;
    const int li0 = tid * chunk;
    if (li0 < n) {
      u128 state = pcg_jump_fast(s_rng[0], A.rng_tab, A.rng_aff, (uint64_t)li0);
      const u128 mult = pcg_mult();
      for (int e = 0; e < chunk && li0 + e < n; ++e) {
        const int li = li0 + e;
        state = state * mult + A.rng_inc;
        const double u = pcg_output(state);
        int jt = -1;
        if (li > 0) {
          const int64_t t = (int64_t)((double)lo + u * (double)(hi - lo)) - lo;
          jt = (int)(t > n - 1 ? n - 1 : (t < 0 ? 0 : t));
          if (atomicCAS(&s0[jt], -1, li) != -1)
            if (atomicCAS(&s1[jt], -1, li) != -1)
              next[li] = (int16_t)atomicExch(&head[jt], li);
        }
        jown[li] = (int16_t)jt;
      }
    }
  }
  __syncthreads();
  CELL_MARK(2);
  // backward walks (see index.hip), entirely in LDS
  for (int li = tid; li < n; li += CELL_THREADS) {
    int e = 0, q = li;
    for (;;) {
      int best = INT32_MAX;
      const int jq = jown[q];
      if (q > e && jq >= 0) best = q;
      const int a = s0[q], b = s1[q];
      if (a > e && a < best) best = a;
      if (b > e && b < best) best = b;
      for (int t = head[q]; t >= 0; t = next[t])
        if (t > e && t < best) best = t;
      if (best == INT32_MAX) break;
      q = (best == q) ? jq : best;
      e = best;
    }
    out[li] = val[q];
  }
  __syncthreads();
  CELL_MARK(3);
  // pairs: positions p with (p - cell_start[cell_idx[cid]]) even and p + 1 in the same segment
  const int64_t cid = s_cid;
  const int lp0 = (int)((lo - s_base) & 1);
  int64_t pj[CELL_MAXPAIR], pk[CELL_MAXPAIR];
  double pprob[CELL_MAXPAIR];
  bool pvalid[CELL_MAXPAIR];
  SD psj[CELL_MAXPAIR], psk[CELL_MAXPAIR];
  double my_min = INFINITY;
  const bool need_r = KERNEL == SDM_KERNEL_GEOMETRIC || KERNEL == SDM_KERNEL_PARAMETERIZED ||
                      KERNEL == SDM_KERNEL_SIMPLE_GEOMETRIC;
  // all gathers of the thread's pairs first (independent loads in flight together) ...
#pragma unroll
  for (int r = 0; r < CELL_MAXPAIR; ++r) {
    const int lp = lp0 + 2 * (tid * CELL_MAXPAIR + r);  // consecutive pair slots per thread
    pvalid[r] = lp + 1 < n && lo + lp < W - 1;
    pj[r] = pk[r] = 0;
    if (pvalid[r]) {
      pj[r] = out[lp];
      pk[r] = out[lp + 1];
    }
  }
#pragma unroll
  for (int r = 0; r < CELL_MAXPAIR; ++r) {
    psj[r].n = psk[r].n = 1; psj[r].m = psk[r].m = psj[r].r = psk[r].r = psj[r].u = psk[r].u = 0;
    if (pvalid[r]) {
      psj[r] = sd_load(cfg, A, pj[r], need_r);
      psk[r] = sd_load(cfg, A, pk[r], need_r);
    }
  }
  // ... then sort within pair, kernel, probability
#pragma unroll
  for (int r = 0; r < CELL_MAXPAIR; ++r) {
    const int lp = lp0 + 2 * (tid * CELL_MAXPAIR + r);
    const int64_t p = lo + lp;
    pprob[r] = 0.0;
    if (pvalid[r]) {
      if (psj[r].n < psk[r].n) {  // sort_within_pair_by_attr
        const int64_t t = pj[r]; pj[r] = pk[r]; pk[r] = t;
        const SD ts = psj[r]; psj[r] = psk[r]; psk[r] = ts;
        out[lp] = (int32_t)pj[r];
        out[lp + 1] = (int32_t)pk[r];
      }
      const double prob = pair_prob_value<KERNEL>(cfg, A, p >> 1, psj[r], psk[r]);
      pprob[r] = prob;
      if (cfg.adaptive && prob != 0) {
        const int64_t prop = psj[r].n / psk[r].n;
        const double t = cfg.dt * (double)prop / prob;
        const double dt_opt = cfg.dt_min > t ? cfg.dt_min : t;
        my_min = dt_opt < my_min ? dt_opt : my_min;
      }
    }
  }
  CELL_MARK(4);
  double scale = 1.0 / (double)cfg.substeps;
  if (cfg.adaptive) {  // workgroup minimum of the optimal sub-step (collisions_methods.py:357-368)
    const double m = wave_min_f64(my_min);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    double bmin = red[0];
    for (int w = 1; w < CELL_THREADS / SDM_WAVE; ++w) bmin = red[w] < bmin ? red[w] : bmin;
    if (tid == 0) A.cell_min[cid] = bmin;  // (k_cells_turn does the per-cell bookkeeping)
    const double l = A.dt_left[cid];
    double todo = cfg.dt_max < l ? cfg.dt_max : l;
    if (bmin < todo) todo = bmin;
    scale = todo / cfg.dt;
  }
  CELL_MARK(5);
  // gamma + update (all lanes take part: wave-aggregated counters).  A thread's pair slots are
  // consecutive, so are their draws: one jump-ahead, then single generator steps.
  // Coalescence only, one extensive attribute (`dense`): the colliding pairs are first listed in
  // LDS (over the hit tables, dead since the walks) and then resolved one per thread - resolved
  // where they are found, every wavefront that holds one pays the divergent update path once per
  // pair slot of its threads (18.6 us per cell against 5.5 us without any collision, measured)
  const bool dense = !BREAKUP && cfg.n_attr == 1;
  __shared__ int s_ncoll;
  double *list_g = (double *)s0;   // [CELL_CAP / 2], over s0 and s1
  int32_t *list_lp = head;         // [CELL_CAP / 2]
  if (dense) {
    if (tid == 0) s_ncoll = 0;
    __syncthreads();
  }
  u128 st = 0, sb = 0;
  {
    const uint64_t dd0 = (uint64_t)(((lo + lp0 + 2 * (int64_t)(tid * CELL_MAXPAIR)) >> 1) -
                                    (lo >> 1));
    st = pcg_jump_fast(s_rng[1], A.rng_tab, A.rng_aff, dd0);
    if (BREAKUP) sb = pcg_jump_fast(s_rng[2], A.rng_tab, A.rng_aff, dd0);
  }
#pragma unroll
  for (int r = 0; r < CELL_MAXPAIR; ++r) {
    const int lp = lp0 + 2 * (tid * CELL_MAXPAIR + r);
    const int64_t d = (lo + lp) >> 1;
    st = st * pcg_mult() + A.rng_inc;
    const double u = pcg_output(st);
    double u_b = 0.0;
    if (BREAKUP) {
      sb = sb * pcg_mult() + A.rng_inc;
      u_b = pcg_output(sb);
    }
    double p = pprob[r];
    if (pvalid[r] && p != 0) { if (cfg.adaptive) p *= scale; else p /= (double)cfg.substeps; }
    if (dense) {  // compute_gamma (collisions_methods.py:560-585) here, the update below
      bool collide = false;
      int64_t nk = 0, gi = 0, gc = 0;
      double g = 0;
      if (pvalid[r]) {
        g = ceil(p - u);
        if (g != 0) {
          collide = true;
          nk = psk[r].n;
          const int64_t prop = psj[r].n / nk;
          gi = (int64_t)g;
          gc = gi < prop ? gi : prop;
          g = (double)gc;
        }
      }
      counter_add(A, CNT_COLLISION, cid, gc * nk, collide);
      counter_add(A, CNT_COLLISION_DEFICIT, cid, (gi - gc) * nk, collide);
      if (collide && g != 0) {
        const int slot = atomicAdd(&s_ncoll, 1);
        list_lp[slot] = lp;
        list_g[slot] = g;
      }
      continue;
    }
    const int died = pair_update_body<BREAKUP>(cfg, A, d, pvalid[r], p, u, u_b, true, 0, pj[r],
                                               pk[r], lo + lp, false, &psj[r], &psk[r]);
    if (died & 1) out[lp] = (int32_t)N;  // the permutation is still in LDS here
    if (died & 2) out[lp + 1] = (int32_t)N;
    if (died) note_deaths(A, blockIdx.x, died);
  }
  __syncthreads();
  if (dense) {
    const int n_coll = s_ncoll;
    for (int base = 0; base < n_coll; base += CELL_THREADS) {
      const int t = base + tid;
      const bool act = t < n_coll;
      int lp = 0;
      double g = 0;
      int64_t j = 0, k = 0;
      SD sj, sk;
      sj.n = sk.n = 1; sj.m = sk.m = sj.r = sk.r = sj.u = sk.u = 0;
      if (act) {
        lp = list_lp[t];
        g = list_g[t];
        j = out[lp];
        k = out[lp + 1];
        sj = sd_load(cfg, A, j, need_r);
        sk = sd_load(cfg, A, k, need_r);
      }
      counter_add(A, CNT_COALESCENCE, cid, (int64_t)(g * (double)sk.n), act);
      const int died = act ? coalesce_known(cfg, A, j, k, g, sj, sk) : 0;
      if (died & 1) out[lp] = (int32_t)N;  // the permutation is still in LDS here
      if (died & 2) out[lp + 1] = (int32_t)N;
      if (died) note_deaths(A, blockIdx.x, died);
    }
    __syncthreads();
  }
  CELL_MARK(6);
  for (int li = tid; li < n; li += CELL_THREADS) X.idx_out[lo + li] = out[li];
#ifdef CELL_PROFILE
  __syncthreads();
  if (blockIdx.x == 7 && tid == 0) {
    const long long t7 = wall_clock64();
    printf("cell n=%d ticks(10ns): load %lld prep %lld events %lld walk %lld gather+prob %lld min %lld update %lld store %lld\n",
           n, cell_t[1] - cell_t[0], cell_t[2] - cell_t[1], cell_t[3] - cell_t[2], cell_t[4] - cell_t[3],
           cell_t[5] - cell_t[4], cell_t[6] - cell_t[5], cell_t[7] - cell_t[6], t7 - cell_t[7]);
  }
#endif
}

// ---- the same sub-step with two workgroups per CU (one extensive attribute) --------------------
// k_cell_step keeps a CU to one workgroup (147 KB of LDS), so its phases - LDS-bound shuffle,
// chip-wide miss-rate-bound gathers, update - never overlap: two co-resident workgroups of 512
// threads gave +23 % at 2048 super-droplets per cell.  To fit two (<= 80 KB each, <= 128 VGPRs):
// 14 B of LDS per position - the two inline hit slots share one word (CAS on the pair), the
// overflow heads are 16-bit (exchange through a CAS on the containing word), the permutation is
// written over the hit words once every walk has finished (results held in registers across a
// barrier) - and no member state is carried in registers into the update: the colliding pairs are
// listed (slot, gamma) and resolved one per thread from the mirror, which is where the per-pair
// counters are formed too.  Pairs are gathered and evaluated three at a time.
#define CELL2_CAP 5632
#define CELL2_THREADS 512
#define CELL2_BATCH 3
#define CELL2_LDS_BYTES (CELL2_CAP * 14)
#define CELL2_PACK 8  // cells per workgroup on the small-cell variant (one wavefront each)
// ... and the same with 384 positions per cell instead of 704 (the reference's own 2-D example has
// 64..128 super-droplets per cell): the per-lane loops, unrolled for the cap, shrink from 11
// positions and 6 pair slots to 6 and 3, and so do the registers - three workgroups per CU
#define CELL2T_CAP (384 * CELL2_PACK)
#define CELL2T_LDS_BYTES (CELL2T_CAP * 14)
// The same kernel with 1024 threads per cell (one workgroup per CU; cells up to 6144): for launches
// in which FEWER cells than the device has CUs are computed - a process of a sharded run that owns
// 128 of the 1024 cells (8 GPUs), say.  Two 512-thread workgroups per CU pay off when there are
// workgroups to pair up; with one cell per CU at most, twice the threads halve every per-thread
// loop of the cell's chain of latencies (events, walks, gathers) instead.
#define CELL2W_CAP 6144
#define CELL2W_THREADS 1024
#define CELL2W_LDS_BYTES (CELL2W_CAP * 14)
// ... and with 256 threads per cell, four workgroups per CU (cells up to 2816): a grid of 1024
// cells of ~1024 is FOUR cells per CU - two rounds of the 512-thread shape, one of this
#define CELL2Q_CAP 2816
#define CELL2Q_THREADS 256
#define CELL2Q_LDS_BYTES (CELL2Q_CAP * 14)

__device__ __forceinline__ int lds_exch16(uint32_t *words, int i, int v) {
  uint32_t *w = words + (i >> 1);
  const int sh = (i & 1) * 16;
  uint32_t old = *(volatile uint32_t *)w;
  for (;;) {
    const uint32_t nw = (old & ~(0xFFFFu << sh)) | ((uint32_t)v << sh);
    const uint32_t prev = atomicCAS(w, old, nw);
    if (prev == old) break;
    old = prev;
  }
  return (int)((old >> sh) & 0xFFFFu);
}

// CPW cells per workgroup (1 or 8): a grid of many small cells - 75 x 75 with 64..128
// super-droplets each is the reference's own 2-D example - would otherwise occupy one 512-thread
// workgroup per cell (5625 workgroups = 11 rounds of the resident 512, each a chain of latencies:
// 170 us per sub-step whatever the cell size).  With CPW = 8 a cell is one wavefront with its own
// slice of LDS (CELL2_CAP / 8 = 704 positions); the barriers stay workgroup-wide (the cells of a
// workgroup move in lockstep), the reductions and counters are per wavefront anyway.
template <int KERNEL, bool BREAKUP, int CPW, int THREADS, bool TINY = false>
__global__ void __launch_bounds__(THREADS, TINY ? 6 : 4)
k_cell_step2(sdm_step_cfg cfg, FusedArgs A, CellArgs X) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef CELL_PROFILE
  __shared__ long long cell_t[10];
  if (threadIdx.x == 0) cell_t[0] = wall_clock64();
#undef CELL_MARK
#define CELL_MARK(k) do { __syncthreads(); if (threadIdx.x == 0) cell_t[(k) + 1] = wall_clock64(); } while (0)
#endif
  constexpr int T = THREADS / CPW;  // threads per cell
  static_assert(!TINY || (CPW == CELL2_PACK && THREADS == CELL2_THREADS), "tiny cells: packed");
  constexpr int CAP = TINY ? CELL2T_CAP / CPW : (THREADS == CELL2_THREADS ? CELL2_CAP :
                       THREADS == CELL2W_THREADS ? CELL2W_CAP : CELL2Q_CAP) / CPW;  // positions per cell
  constexpr int MAXPOS = CAP / T, MAXPAIR = (CAP / 2 + T - 1) / T;
  static_assert(T % SDM_WAVE == 0 && CAP % 8 == 0 && CAP % T == 0, "cell slices");
  static_assert(MAXPAIR % CELL2_BATCH == 0, "pairs are taken in whole batches");
  static_assert(CPW == 1 || THREADS == CELL2_THREADS, "packed cells: the 512-thread shape");
  const int sub = threadIdx.x / T, tid = threadIdx.x % T;  // which cell of the workgroup, lane in it
  char *cmem = smem + (size_t)sub * CAP * 14;
  uint32_t *hits = (uint32_t *)cmem;                 // [CAP] two 16-bit hit slots, 0xFFFF = free
  int32_t *out = (int32_t *)cmem;                    // ... later the permuted ids
  int32_t *val = (int32_t *)(cmem + CAP * 4);        // [CAP] ids before the shuffle
  double *list_g = (double *)val;                    // ... later gamma of the colliding pairs
  uint16_t *head = (uint16_t *)(cmem + CAP * 8);     // [CAP] overflow list heads
  int32_t *list_lp = (int32_t *)head;                // ... later their pair slots
  int16_t *jown = (int16_t *)(cmem + CAP * 10);      // [CAP] own target
  uint16_t *next = (uint16_t *)(cmem + CAP * 12);    // [CAP] overflow links
  double *list_ub = (double *)jown;                  // ... later (breakup) their second draws
  __shared__ double red[THREADS / SDM_WAVE];
  __shared__ int64_t s_cid_[CPW], s_base_[CPW];
  __shared__ u128 s_rng_[CPW][3];
  __shared__ int s_ncoll_[CPW];
  int64_t &s_cid = s_cid_[sub], &s_base = s_base_[sub];
  u128 *s_rng = s_rng_[sub];
  int &s_ncoll = s_ncoll_[sub];
  const int64_t C = cfg.n_cell, N = cfg.n_sd;
  const int64_t n_cell_groups = (C + CPW - 1) / CPW;
  if (X.gate && X.gate[0] == 0) return;
  if ((int64_t)blockIdx.x >= n_cell_groups) {  // dead tail [cell_start[C], N) is carried over unchanged
    const int64_t from = A.cell_start[C];
    for (int64_t i = from + ((int64_t)blockIdx.x - n_cell_groups) * THREADS + threadIdx.x;
         i < N; i += (int64_t)X.n_tail_blocks * THREADS)
      X.idx_out[i] = X.idx_in[i];
    return;
  }
  // a cell that has nothing to do here (beyond the grid, empty, another process's, too large)
  // takes part in the barriers below with n = 0
  const int64_t cell = (int64_t)blockIdx.x * CPW + sub;
  int64_t lo = 0, hi = 0;
  if (cell < C) { lo = A.cell_start[cell]; hi = A.cell_start[cell + 1]; }
  int n = (int)(hi - lo);
  if (n > 0 && A.cell_owned &&
      !(A.seg_owned ? A.seg_owned[cell] : A.cell_owned[A.cell_id[X.idx_in[lo]]])) {
    // another process's cell
    if (X.copy_others)
      for (int li = tid; li < n; li += T) X.idx_out[lo + li] = X.idx_in[lo + li];
    n = 0;
  }
  if (n > CAP) {  // never taken: the host enables this path only below the cap
    if (tid == 0) A.ctl[7] = 1;
    for (int li = tid; li < n; li += T) X.idx_out[lo + li] = X.idx_in[lo + li];
    n = 0;
  }
  // (one cell per workgroup and nothing to do: leave - a rank of a sharded run launches a
  // workgroup per segment and computes an eighth of them)
  if (CPW == 1 && n == 0) return;
  if (CPW > 1 && A.cell_owned && __syncthreads_or(n > 0) == 0) return;  // (none of its cells)
  const int64_t W = A.ctl[CTL_WORK];
  for (int li = tid; li < n; li += T) {
    val[li] = (int32_t)X.idx_in[lo + li];
    hits[li] = 0xFFFFFFFFu;
    head[li] = 0xFFFFu;
  }
  if (tid == 0) { s_ncoll = 0; s_cid = 0; s_base = 0; }
  __syncthreads();
  if (n > 0) {
    // four short serial jobs - the cell's id and pair parity (two or three dependent look-ups), the
    // jump-aheads of the streams - by four threads of DIFFERENT wavefronts where the cell has
    // several (in one wavefront the divergent branches run one after the other: 5 us of the
    // cell's chain)
    constexpr int ROLE = T >= 4 * SDM_WAVE ? SDM_WAVE : 1;
    if (tid == 0) {
      // (seg_cid: written by k_cells_turn after a sort - one look-up instead of two dependent ones)
      s_cid = A.seg_cid ? (int64_t)A.seg_cid[cell] : A.cell_id[val[0]];
      s_base = A.cell_start[A.cell_idx[s_cid]];
    }
    if (tid == ROLE) s_rng[0] = pcg_jump_fast(X.s_u01, A.rng_tab, A.rng_aff, (uint64_t)lo);
    if (tid == 2 * ROLE)
      s_rng[1] = pcg_jump_fast(A.s_rand, A.rng_tab, A.rng_aff, (uint64_t)(lo >> 1));
    if (BREAKUP && tid == 3 * ROLE)
      s_rng[2] = pcg_jump_fast(A.s_rand_b, A.rng_tab, A.rng_aff, (uint64_t)(lo >> 1));
  }
  __syncthreads();
  CELL_MARK(0);
  // The normalisation factors of this thread's pair slots (collisions_methods.py:633-662: that of
  // the cell of RAW super-droplet d for slot d - three dependent look-ups and three divisions per
  // slot) depend on nothing the shuffle produces: requested here, they arrive while the LDS phases
  // run instead of sitting on the critical path of the probability phase (20 of a cell's 52 us).
  // (not with the parameterized kernels: their efficiency polynomial leaves no registers to
  // carry six more doubles through the shuffle)
#ifdef SDM_NO_NORM_AHEAD  // (A/B build)
  constexpr bool NORM_AHEAD = false;
#else
  constexpr bool NORM_AHEAD = KERNEL != SDM_KERNEL_PARAMETERIZED;
#endif
  const int lp0 = (int)((lo - s_base) & 1);
  // a thread's pair slots are consecutive (so are their draws: one jump-ahead, then single
  // generator steps) - `per` of them, the cell's pairs spread evenly over its threads (a cell of
  // 4096 keeps all 512 threads busy with 4 slots each, where MAXPAIR slots per thread, sized for
  // the largest cell, left a third of them without any)
  const int per = (((n - lp0) >> 1) + T - 1) / T;
  double pnorm[NORM_AHEAD ? MAXPAIR : 1];
  if (NORM_AHEAD) {
#pragma unroll
    for (int r = 0; r < MAXPAIR; ++r) {
      const int lp = lp0 + 2 * (tid * per + r);
      const bool valid = r < per && lp + 1 < n && lo + lp < W - 1;
      pnorm[r] = valid ? norm_factor_of(cfg, A.cell_start,
                                        A.cell_idx[A.cell_id_raw[(lo + lp) >> 1]])
                       : 0.0;
    }
  }
  // shuffle_local events of this cell (index_methods.py:35-41): consecutive positions per thread
  {
    const int chunk = (n + T - 1) / T;
    const int li0 = tid * chunk;
    if (li0 < n) {
      u128 state = pcg_jump_fast(s_rng[0], A.rng_tab, A.rng_aff, (uint64_t)li0);
      const u128 mult = pcg_mult();
      for (int e = 0; e < chunk && li0 + e < n; ++e) {
        const int li = li0 + e;
        state = state * mult + A.rng_inc;
        const double u = pcg_output(state);
        int jt = -1;
        if (li > 0) {
          const int64_t t = (int64_t)((double)lo + u * (double)(hi - lo)) - lo;
          jt = (int)(t > n - 1 ? n - 1 : (t < 0 ? 0 : t));
          uint32_t old = *(volatile uint32_t *)&hits[jt];
          for (;;) {
            uint32_t nw;
            if ((old & 0xFFFFu) == 0xFFFFu) nw = (old & 0xFFFF0000u) | (uint32_t)li;
            else if ((old >> 16) == 0xFFFFu) nw = (old & 0xFFFFu) | ((uint32_t)li << 16);
            else { next[li] = (uint16_t)lds_exch16((uint32_t *)head, jt, li); break; }
            const uint32_t prev = atomicCAS(&hits[jt], old, nw);
            if (prev == old) break;
            old = prev;
          }
        }
        jown[li] = (int16_t)jt;
      }
    }
  }
  __syncthreads();
  CELL_MARK(1);
  // backward walks (see index.hip), entirely in LDS; results stay in registers until every walk
  // is through with the hit words.  (Advancing a thread's walks in lockstep - all their reads in
  // flight together - and issuing the hit-slot claims of the events phase together were built and
  // measured in round 3: both slower, 14 against 10.8 us and 13 against 7.2 us per cell; sixteen
  // wavefronts per CU hide the LDS latency already, the extra instructions only add to the issue
  // load.  profiles/r03_cell_lockstep_walks_experiment.patch.  Likewise one 8-byte record per
  // look-up - hit slots and an overflow flag | own target and link - instead of three arrays:
  // parity-green, 126.6 against 122.7 us; profiles/r03_cell_packed_lds_record_experiment.patch)
  int32_t walked[MAXPOS];
#pragma unroll
  for (int w = 0; w < MAXPOS; ++w) {
    const int li = tid + w * T;
    walked[w] = 0;
    if (li < n) {
      int e = 0, q = li;
      for (;;) {
        int best = INT32_MAX;
        const int jq = jown[q];
        if (q > e && jq >= 0) best = q;
        const uint32_t h = hits[q];
        const int a = (int)(h & 0xFFFFu), b = (int)(h >> 16);
        if (a != 0xFFFF && a > e && a < best) best = a;
        if (b != 0xFFFF && b > e && b < best) best = b;
        for (int t = head[q]; t != 0xFFFF; t = next[t])
          if (t > e && t < best) best = t;
        if (best == INT32_MAX) break;
        q = (best == q) ? jq : best;
        e = best;
      }
      walked[w] = val[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < MAXPOS; ++w) {
    const int li = tid + w * T;
    if (li < n) out[li] = walked[w];
  }
  __syncthreads();
  CELL_MARK(2);
  // pairs: positions p with (p - cell_start[cell_idx[cid]]) even and p + 1 in the same segment
  const int64_t cid = s_cid;
  double pprob[MAXPAIR];
  double my_min = INFINITY;
  const bool need_r = KERNEL == SDM_KERNEL_GEOMETRIC || KERNEL == SDM_KERNEL_PARAMETERIZED ||
                      KERNEL == SDM_KERNEL_SIMPLE_GEOMETRIC;
#pragma unroll
  for (int b0 = 0; b0 < MAXPAIR; b0 += CELL2_BATCH) {
    int64_t pj[CELL2_BATCH], pk[CELL2_BATCH];
    bool pvalid[CELL2_BATCH];
    SD psj[CELL2_BATCH], psk[CELL2_BATCH];
#pragma unroll
    for (int r = 0; r < CELL2_BATCH; ++r) {
      const int lp = lp0 + 2 * (tid * per + b0 + r);  // consecutive pair slots per thread
      pvalid[r] = b0 + r < per && lp + 1 < n && lo + lp < W - 1;
      pj[r] = pk[r] = 0;
      if (pvalid[r]) {
        pj[r] = out[lp];
        pk[r] = out[lp + 1];
      }
    }
#pragma unroll
    for (int r = 0; r < CELL2_BATCH; ++r) {
      psj[r].n = psk[r].n = 1;
      psj[r].m = psk[r].m = psj[r].r = psk[r].r = psj[r].u = psk[r].u = 0;
      if (pvalid[r]) {
        psj[r] = sd_load(cfg, A, pj[r], need_r);
        psk[r] = sd_load(cfg, A, pk[r], need_r);
      }
    }
#pragma unroll
    for (int r = 0; r < CELL2_BATCH; ++r) {
      const int lp = lp0 + 2 * (tid * per + b0 + r);
      const int64_t p = lo + lp;
      pprob[b0 + r] = 0.0;
      if (pvalid[r]) {
        if (psj[r].n < psk[r].n) {  // sort_within_pair_by_attr
          const SD ts = psj[r]; psj[r] = psk[r]; psk[r] = ts;
          out[lp] = (int32_t)pk[r];
          out[lp + 1] = (int32_t)pj[r];
        }
        const double prob = pair_prob_value<KERNEL>(cfg, A, p >> 1, psj[r], psk[r],
                                                    NORM_AHEAD ? &pnorm[b0 + r] : nullptr);
        pprob[b0 + r] = prob;
        if (cfg.adaptive && prob != 0) {
          const int64_t prop = psj[r].n / psk[r].n;
          const double t = cfg.dt * (double)prop / prob;
          const double dt_opt = cfg.dt_min > t ? cfg.dt_min : t;
          my_min = dt_opt < my_min ? dt_opt : my_min;
        }
      }
    }
  }
  CELL_MARK(3);
  double scale = 1.0 / (double)cfg.substeps;
  if (cfg.adaptive) {  // workgroup minimum of the optimal sub-step (collisions_methods.py:357-368)
    const double m = wave_min_f64(my_min);
    if ((tid & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const int w0 = sub * (T / SDM_WAVE);  // this cell's wavefronts
    double bmin = red[w0];
    for (int w = 1; w < T / SDM_WAVE; ++w) bmin = red[w0 + w] < bmin ? red[w0 + w] : bmin;
    if (tid == 0 && n > 0) A.cell_min[cid] = bmin;  // (k_cells_turn does the per-cell bookkeeping)
    const double l = A.dt_left[cid];
    double todo = cfg.dt_max < l ? cfg.dt_max : l;
    if (bmin < todo) todo = bmin;
    scale = todo / cfg.dt;
  }
  CELL_MARK(4);
  // gamma (collisions_methods.py:560): the pairs that collide are listed, over val / head
  {
    const uint64_t dd0 = (uint64_t)(((lo + lp0 + 2 * (int64_t)(tid * per)) >> 1) - (lo >> 1));
    u128 st = pcg_jump_fast(s_rng[1], A.rng_tab, A.rng_aff, dd0), sb = 0;
    if (BREAKUP) sb = pcg_jump_fast(s_rng[2], A.rng_tab, A.rng_aff, dd0);
#pragma unroll
    for (int r = 0; r < MAXPAIR; ++r) {
      const int lp = lp0 + 2 * (tid * per + r);
      st = st * pcg_mult() + A.rng_inc;
      const double u = pcg_output(st);
      double u_b = 0.0;
      if (BREAKUP) {
        sb = sb * pcg_mult() + A.rng_inc;
        u_b = pcg_output(sb);
      }
      double p = pprob[r];
      const bool valid = r < per && lp + 1 < n && lo + lp < W - 1;
      if (valid && p != 0) { if (cfg.adaptive) p *= scale; else p /= (double)cfg.substeps; }
      const double g = valid ? ceil(p - u) : 0.0;
      if (g != 0) {
        const int slot = atomicAdd(&s_ncoll, 1);
        list_lp[slot] = lp;
        list_g[slot] = g;
        if (BREAKUP) list_ub[slot] = u_b;
      }
    }
  }
  __syncthreads();
  CELL_MARK(5);
  // update: one colliding pair per thread, state from the mirror (compute_gamma's clamp and
  // counters :566-585, coalescence :44-59)
  const int n_coll = s_ncoll;
  for (int base = 0; base < n_coll; base += T) {
    const int t = base + tid;
    const bool act = t < n_coll;
    int lp = 0;
    double g = 0;
    int64_t j = 0, k = 0, gi = 0, gc = 0;
    SD sj, sk;
    sj.n = sk.n = 1; sj.m = sk.m = sj.r = sk.r = sj.u = sk.u = 0;
    if (act) {
      lp = list_lp[t];
      g = list_g[t];
      j = out[lp];
      k = out[lp + 1];
      sj = sd_load(cfg, A, j, need_r);
      sk = sd_load(cfg, A, k, need_r);
      const int64_t prop = sj.n / sk.n;
      gi = (int64_t)g;
      gc = gi < prop ? gi : prop;
      g = (double)gc;
    }
    counter_add(A, CNT_COLLISION, cid, gc * sk.n, act);
    counter_add(A, CNT_COLLISION_DEFICIT, cid, (gi - gc) * sk.n, act);
    const bool coal = act && g != 0;
    if (BREAKUP) {
      // bounce / coalescence / breakup is decided and applied by k_resolve_dense: list the pair
      const unsigned long long m = __ballot(coal);
      if (m != 0) {
        const int lane = lane_id(), leader = __ffsll((long long)m) - 1;
        const int64_t l = blockIdx.x % (unsigned)A.list_nl;
        unsigned long long at = 0;
        if (lane == leader) at = atomicAdd(A.list_count + l * SDM_CNT_STRIDE,
                                           (unsigned long long)__popcll(m));
        at = __shfl((long long)at, leader, 64);
        if (coal) {
          Collided c;
          c.j = j; c.k = k; c.cid = cid; c.pos = lo + lp; c.g = g; c.u_b = list_ub[t];
          A.list[l * A.list_cap + at + __popcll(m & ((1ull << lane) - 1))] = c;
        }
      }
      continue;
    }
    counter_add(A, CNT_COALESCENCE, cid, (int64_t)(g * (double)sk.n), coal);
    const int died = coal ? coalesce_known(cfg, A, j, k, g, sj, sk) : 0;
    if (died & 1) out[lp] = (int32_t)N;  // the permutation is still in LDS here
    if (died & 2) out[lp + 1] = (int32_t)N;
    if (died) note_deaths(A, cell, died);
  }
  __syncthreads();
  CELL_MARK(6);
  for (int li = tid; li < n; li += T) X.idx_out[lo + li] = out[li];
#ifdef CELL_PROFILE
  __syncthreads();
  // (the workgroup of cell 7: whichever position the cell order gives it, owned by rank 0 of a
  // sharded run)
  if (n > 0 && s_cid_[0] == 7 && threadIdx.x == 0) {
    const long long t_end = wall_clock64();
    printf("cell2 threads=%d n=%d colliding=%d ticks(10ns): load+init %lld events %lld walks %lld gather+prob %lld "
           "min %lld gamma %lld update %lld store %lld\n", THREADS, n, s_ncoll_[0], cell_t[1] - cell_t[0],
           cell_t[2] - cell_t[1], cell_t[3] - cell_t[2], cell_t[4] - cell_t[3],
           cell_t[5] - cell_t[4], cell_t[6] - cell_t[5], cell_t[7] - cell_t[6], t_end - cell_t[7]);
  }
#endif
}

// largest cell of a sorted state -> ctl[6].  The one place a caller's claim "sorted" is checked,
// for what four loads can tell: cell_start must be non-decreasing and span exactly the live
// super-droplets.  A cell_start left over from another state (a restored snapshot that forgot it,
// say) puts droplets into other cells' segments; the sub-step loop then never ends, because the
// cells whose time is charged are not the cells that are waited for -> ctl[7] |= 4, the host
// returns SDM_E_ARG before anything is computed
// `owned` / `n_owned` (sharded mode, else NULL): how many cells this process computes -> *n_owned
__global__ void __launch_bounds__(SDM_BLOCK)
k_max_cell(const int64_t *__restrict__ cell_start, int64_t n_cell, int64_t *ctl,
           const uint8_t *__restrict__ owned, int64_t *n_owned) {
  const int64_t c = TID();
  const int64_t sz = c < n_cell ? cell_start[c + 1] - cell_start[c] : 0;
  if (owned) {
    const unsigned long long mine = __ballot(c < n_cell && owned[c] != 0);
    if (mine && lane_id() == 0)
      atomicAdd((unsigned long long *)n_owned, (unsigned long long)__popcll(mine));
  }
  // (an empty state is never sorted - the sort's kernels exit at once - and nothing reads its
  // cell_start)
  if (ctl[CTL_VALID] != 0 &&
      (sz < 0 || (c == 0 && ctl[CTL_SORTED] != 0 &&
                  (cell_start[0] != 0 || cell_start[n_cell] != ctl[CTL_VALID]))))
    atomicOr((unsigned long long *)&ctl[7], 4ull);
  int64_t m = sz;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t t = __shfl_xor((long long)m, o, 64);
    m = t > m ? t : m;
  }
  if (lane_id() == 0) atomicMax((long long *)&ctl[6], (long long)m);
}

// ---- breakup: dense resolution of the listed colliding pairs ----------------------------------
__global__ void __launch_bounds__(SDM_BLOCK) k_resolve_dense(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t l = blockIdx.x % (unsigned)A.list_nl, chunk = blockIdx.x / (unsigned)A.list_nl;
  if (blockIdx.x == 0 && threadIdx.x < LIST_NL) A.list_count_next[threadIdx.x * SDM_CNT_STRIDE] = 0;
  const int64_t n = (int64_t)A.list_count[l * SDM_CNT_STRIDE];
  if (chunk * SDM_BLOCK >= n) return;
  const int64_t t = chunk * SDM_BLOCK + threadIdx.x;
  const bool active = t < n;
  Collided c;
  c.j = c.k = c.cid = c.pos = 0; c.g = 0; c.u_b = 0;
  if (active) c = A.list[l * A.list_cap + t];
  const int died = resolve_collision<true>(cfg, A, active, c.j, c.k, c.cid, c.g, c.u_b);
  if (died) flag_dead(cfg, A.idx, c.pos, died, &A);
}

// ---- control-word kernels -------------------------------------------------------------------
// conditional counting sort (multi-cell): the sort cores read the length from gate_len (0
// disables them); whether to run is decided on the device by ctl[CTL_SORTED].
// gate_len[1]: "the sort runs" (k_sort_commit then copies the result in and marks the state sorted)
__global__ void k_sort_gate(int64_t *ctl, int64_t *gate_len) {
  gate_len[0] = ctl[CTL_SORTED] ? 0 : ctl[CTL_WORK];
  gate_len[1] = (ctl[CTL_SORTED] || ctl[CTL_WORK] == 0) ? 0 : 1;  // (nothing to sort: no commit)
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_sort_commit(int64_t *__restrict__ idx, const int64_t *__restrict__ sorted_buf,
              int64_t *__restrict__ cell_start, const int64_t *__restrict__ cs_tmp,
              int64_t n_cell, int64_t *ctl, const int64_t *__restrict__ gate_len) {
  const int64_t n = gate_len[0];
  if (gate_len[1] == 0) return;
  const int64_t i = TID();
  if (i < n) idx[i] = sorted_buf[i];
  if (i <= n_cell) cell_start[i] = cs_tmp[i];
  if (i == 0) ctl[CTL_SORTED] = 1;
}

__global__ void k_mark_unsorted(int64_t *ctl) { ctl[CTL_SORTED] = 0; }

__global__ void __launch_bounds__(SDM_BLOCK) k_nm_init(sdm_step_cfg cfg, FusedArgs A) {
  const int64_t i = TID();
  if (i < cfg.n_sd) sd_refresh(cfg, A, i);
}

// single cell: the sort is the identity, only cell_start = {0, length} has to be right
__global__ void k_single_cell_init(int64_t *ctl, int64_t *cell_start) {
  cell_start[0] = 0;
  cell_start[1] = ctl[CTL_WORK];
  ctl[CTL_SORTED] = 1;
}

// collision.py:185-187 for one cell: working length = whole cell while dt_left > 0
__global__ void k_set_work(int64_t *ctl, const int64_t *end, int64_t *box, int64_t seq) {
  ctl[CTL_WORK] = end[0];
  publish_ctl(ctl, box, seq, end[0]);
}
__global__ void k_reset_work(int64_t *ctl) { ctl[CTL_WORK] = ctl[CTL_VALID]; }
// multi-cell: reset_working_length + reset_cell_idx (identity; un-sorts) in one launch
// ---- sharded mode: what crosses process boundaries ---------------------------------------------
// Invariant of a process's permutation: the segments of the cells it owns are exact; every other
// segment holds as many ids as the cell has members, each of them an id of THAT cell (which ones,
// and in which order, is the owner's business).  That is all the replicated compaction and the
// stable counting sort need to put the owned segments in the reference's order
// (collisions_methods.py:664-697): positions and cell sizes are global, ids only travel with
// their cell.
//
// After the per-cell bookkeeping of a sub-step: x[c] = dt_left[c] of the cells this process owns
// (0 elsewhere), x[C] = number of super-droplets that died in them, x[C + 1 + r] = the same,
// filed under the process's rank r; summed over the processes by the caller's exchange, then
// written back for every cell.
// `dead`: positions of the flagged entries of the owned segments (a super-droplet that died was
// flagged - and counted, FusedArgs::n_dead - where it sits by the kernel that updated it:
// k_cell_step*, k_resolve_dense, k_pair_*).  Launched only once the exchange of the counts has
// shown that a super-droplet died somewhere (rare): the sub-steps in which nobody does pay nothing
__global__ void __launch_bounds__(SDM_BLOCK)
k_shard_dead_list(FusedArgs A, const int64_t *__restrict__ idx, int64_t n_cell, int64_t n_sd,
                  int64_t *__restrict__ dead, unsigned long long *__restrict__ n_dead) {
  const int64_t lo = A.cell_start[blockIdx.x], hi = A.cell_start[blockIdx.x + 1];
  if (hi == lo) return;
  // (a flagged entry can only have been written by the owner: this process)
  const int64_t first = idx[lo];
  if (first < n_sd && !A.cell_owned[A.cell_id[first]]) return;
  for (int64_t base = lo; base < hi; base += SDM_BLOCK) {
    const int64_t i = base + threadIdx.x;
    const bool is_dead = i < hi && idx[i] >= n_sd;
    const unsigned long long m = __ballot(is_dead);
    if (m == 0) continue;
    const int lane = lane_id(), leader = __ffsll((long long)m) - 1;
    unsigned long long at = 0;
    if (lane == leader) at = atomicAdd(n_dead, (unsigned long long)__popcll(m));
    at = __shfl((long long)at, leader, 64);
    if (is_dead) dead[at + __popcll(m & ((1ull << lane) - 1))] = i;
  }
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_shard_pack(FusedArgs A, int64_t n_cell, int adaptive, double *__restrict__ x,
             const unsigned long long *__restrict__ n_dead, int rank, int world) {
  const int64_t c = TID();
  if (c < n_cell) x[c] = (adaptive && A.cell_owned[c]) ? A.dt_left[c] : 0.0;
  if (c == n_cell) x[c] = (double)n_dead[0];
  if (c > n_cell && c <= n_cell + world) x[c] = (c - n_cell - 1 == rank) ? (double)n_dead[0] : 0.0;
}
// adaptive per-cell route (k_cells_turn's buffers): neg[k] = minus the number of super-droplets that
// died in segment k, from every process (all-reduce MIN) -> off[k] = where segment k's dead go in
// the list of positions (segment order: the same on every process), *total
__global__ void __launch_bounds__(1024)
k_shard_dead_offsets(const double *__restrict__ neg, int64_t n, int64_t *__restrict__ off,
                     int64_t *__restrict__ total) {
  __shared__ int64_t sm[1024];
  const int64_t per = (n + 1023) / 1024, c0 = (int64_t)threadIdx.x * per;
  const int64_t c1 = c0 + per < n ? c0 + per : n;
  int64_t sum = 0;
  for (int64_t c = c0; c < c1; ++c) sum += (int64_t)(-neg[c]);
  sm[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int64_t t = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t run = sm[threadIdx.x] - sum;
  for (int64_t c = c0; c < c1; ++c) {
    off[c] = run;
    run += (int64_t)(-neg[c]);
  }
  if (threadIdx.x == 1023) *total = sm[1023];
}
// one workgroup per segment in which a super-droplet died: the positions of its flagged entries,
// at the segment's place in the list.  Only the owner's permutation carries this sub-step's flags
// (all earlier ones were compacted away), so the other processes find none and write nothing
__global__ void __launch_bounds__(SDM_BLOCK)
k_shard_dead_list_by_segment(const int64_t *__restrict__ cell_start, const double *__restrict__ neg,
                             const int64_t *__restrict__ off, const int64_t *__restrict__ idx,
                             int64_t n_sd, int64_t *__restrict__ out) {
  if (neg[blockIdx.x] == 0) return;
  __shared__ int found;
  if (threadIdx.x == 0) found = 0;
  __syncthreads();
  const int64_t lo = cell_start[blockIdx.x], hi = cell_start[blockIdx.x + 1];
  const int64_t base = off[blockIdx.x];
  for (int64_t i = lo + threadIdx.x; i < hi; i += SDM_BLOCK)
    if (idx[i] >= n_sd) out[base + atomicAdd(&found, 1)] = i;
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_shard_unpack(FusedArgs A, int64_t n_cell, int adaptive, const double *__restrict__ x) {
  const int64_t c = TID();
  if (c < n_cell && adaptive) A.dt_left[c] = x[c];
  if (c == n_cell && x[c] > 0) A.ctl[CTL_HEALTHY] = 0;
}
// this process's dead positions into its slice of the (zeroed) exchange buffer
__global__ void __launch_bounds__(SDM_BLOCK)
k_shard_dead_place(const int64_t *__restrict__ dead, int64_t n, int64_t *__restrict__ out) {
  const int64_t i = TID();
  if (i < n) out[i] = dead[i];
}
// every process flags the positions at which a super-droplet died anywhere (in a segment of
// another process this removes SOME member of that cell from the local copy: see the invariant)
__global__ void __launch_bounds__(SDM_BLOCK)
k_shard_flag(int64_t *__restrict__ idx, const int64_t *__restrict__ dead, int64_t n,
             int64_t n_sd) {
  const int64_t i = TID();
  if (i < n) idx[dead[i]] = n_sd;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_step_close(int64_t *ctl, int64_t *__restrict__ cell_idx, int64_t n_cell, int64_t *gate_len,
             int64_t *__restrict__ seg_size) {
  const int64_t c = TID();
  if (c < n_cell) {
    cell_idx[c] = c;
    seg_size[c] = 0;
  }
  if (c == 0) {
    ctl[CTL_WORK] = ctl[CTL_VALID];
    ctl[CTL_SORTED] = 0;
    gate_len[0] = ctl[CTL_VALID];  // = what k_sort_gate would find for the sort that follows
    gate_len[1] = 1;
  }
}

// (z0, z1: words cleared on the way - the largest-cell word and the owned-cell count that
// k_max_cell accumulates into; NULL: none)
__global__ void __launch_bounds__(SDM_BLOCK) k_fill_f64(double *p, double v, int64_t n,
                                                        int64_t *z0 = nullptr,
                                                        int64_t *z1 = nullptr) {
  const int64_t i = TID();
  if (i < n) p[i] = v;
  if (i == 0) {
    if (z0) *z0 = 0;
    if (z1) *z1 = 0;
  }
}

// ---------------------------------------------------------------------------------------------
struct FusedScratch {
  double *prob, *dt_todo, *cell_min, *block_min;
  Collided *list;
  unsigned long long *list_count;
  int64_t flat_list_cap;  // capacity of each of the LIST_NL lists of the flat pair kernels
  uint8_t *pair_off;
  int32_t *pair_cid;
  int64_t *sorted_buf, *cs_tmp, *gate_len, *cctl, *end2, *seg_size, *seg_src;
  uint8_t *seg_owned;
  int32_t *seg_cid;
  int64_t *resort_plan;  // (index.hip: sdm_resort_after_compaction_async)
  char *shuffle, *sort, *compact;
  size_t total;
};

static FusedScratch layout(char *base, const sdm_step_cfg *cfg) {
  FusedScratch S;
  Carver cv(base);
  const int64_t N = cfg->n_sd, P = N / 2 > 0 ? N / 2 : 1, C = cfg->n_cell;
  const bool split = cfg->adaptive != 0;  // prob etc. cross a kernel boundary only then
  S.prob = cv.take<double>(split ? P : 1);
  S.dt_todo = cv.take<double>(C);
  // flat pair kernels: workgroups of SDM_BLOCK pair slots, list l fed by every LIST_NL-th of them
  S.flat_list_cap = (grid_for((N + 1) / 2) + LIST_NL - 1) / LIST_NL * SDM_BLOCK;
  S.list = cv.take<Collided>(cfg->enable_breakup ? std::max<int64_t>(P, LIST_NL * S.flat_list_cap)
                                                 : 1);
  S.list_count = cv.take<unsigned long long>(2 * LIST_NL * SDM_CNT_STRIDE);
  S.cell_min = cv.take<double>(4 * C);  // (k_cells_turn: two buffers of minima + deaths)
  S.block_min = cv.take<double>(grid_for((cfg->n_sd + 1) / 2) + 1);
  S.pair_off = cv.take<uint8_t>(split ? P : 1);
  S.pair_cid = cv.take<int32_t>(split ? P : 1);
  S.sorted_buf = cv.take<int64_t>(C > 1 ? N : 1);
  S.cs_tmp = cv.take<int64_t>(C + 1);
  S.seg_size = cv.take<int64_t>(C);
  S.seg_src = cv.take<int64_t>(C);
  S.seg_owned = cv.take<uint8_t>(C);
  S.seg_cid = cv.take<int32_t>(C);
  S.gate_len = cv.take<int64_t>(4);
  S.cctl = cv.take<int64_t>(8);
  S.resort_plan = cv.take<int64_t>(8);
  S.end2 = cv.take<int64_t>(8);  // (words 2-3: k_cells_turn's gates; sharded mode: 4 deaths of a sub-step, 5 owned cells, 6 listed)
  S.shuffle = base + cv.off;
  cv.off += carve_size(sdm_shuffle_scratch(N));
  S.sort = base + cv.off;
  cv.off += carve_size(sdm_sort_scratch(N, C));
  S.compact = base + cv.off;
  cv.off += carve_size(sdm_compact_scratch(N));
  S.total = cv.off;
  return S;
}

// 32-B mirror records {multiplicity, mass, radius, velocity} where the kernel or the breakup
// parts need radius / velocity, 16-B {multiplicity, mass} otherwise
static bool mirror_is_wide(const sdm_step_cfg *cfg) {
  return cfg->kernel == SDM_KERNEL_GEOMETRIC || cfg->kernel == SDM_KERNEL_PARAMETERIZED ||
         cfg->kernel == SDM_KERNEL_SIMPLE_GEOMETRIC ||
         (cfg->enable_breakup && (cfg->ec != SDM_EC_CONST || cfg->frag == SDM_FRAG_STRAUB2010 ||
                                  cfg->frag == SDM_FRAG_LOWLIST1982));
}

// `known`: what the host knows about ctl[CTL_SORTED] (1 sorted, 0 unsorted, -1 unknown);
// afterwards the device state is sorted in any case
// `gated`: gate_len was already written (k_step_close)
static int cond_sort(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st, int64_t *idx,
                     const FusedScratch &S, int *known, bool gated = false) {
  if (cfg->n_cell == 1) return SDM_OK;  // identity; cell_start kept right by the compaction
  if (*known == 1) return SDM_OK;
  *known = 1;
  PhaseScope ph(ctx, SDM_PHASE_SORT);
  if (!gated) {
    hipLaunchKernelGGL(k_sort_gate, dim3(1), dim3(1), 0, ctx->stream, st->ctl, S.gate_len);
    LAUNCH_CHECK();
  }
  int rc = sdm_counting_sort_async(ctx, S.sort, S.sorted_buf, idx, st->cell_id, st->cell_idx,
                                   S.gate_len, cfg->n_sd, S.cs_tmp, cfg->n_cell);
  if (rc) return rc;
  const int64_t n = cfg->n_sd > cfg->n_cell + 1 ? cfg->n_sd : cfg->n_cell + 1;
  hipLaunchKernelGGL(k_sort_commit, dim3(grid_for(n)), dim3(SDM_BLOCK), 0, ctx->stream, idx,
                     S.sorted_buf, st->cell_start, S.cs_tmp, cfg->n_cell, st->ctl, S.gate_len);
  LAUNCH_CHECK();
  return SDM_OK;
}

// kernel specialisations: collision kernel kind x breakup on/off (keeps the coalescence-only
// kernels free of the breakup code's registers)
#define DISPATCH_PAIR(KERN, GRID)                                                              \
  do {                                                                                         \
    const bool brk__ = cfg->enable_breakup != 0;                                               \
    switch (cfg->kernel) {                                                                     \
      case SDM_KERNEL_GOLOVIN:                                                                 \
        if (brk__) hipLaunchKernelGGL((KERN<SDM_KERNEL_GOLOVIN, true>), GRID, blk, 0, s, *cfg, A); \
        else hipLaunchKernelGGL((KERN<SDM_KERNEL_GOLOVIN, false>), GRID, blk, 0, s, *cfg, A);  \
        break;                                                                                 \
      case SDM_KERNEL_GEOMETRIC:                                                               \
        if (brk__) hipLaunchKernelGGL((KERN<SDM_KERNEL_GEOMETRIC, true>), GRID, blk, 0, s, *cfg, A); \
        else hipLaunchKernelGGL((KERN<SDM_KERNEL_GEOMETRIC, false>), GRID, blk, 0, s, *cfg, A); \
        break;                                                                                 \
      case SDM_KERNEL_PARAMETERIZED:                                                               \
        if (brk__) hipLaunchKernelGGL((KERN<SDM_KERNEL_PARAMETERIZED, true>), GRID, blk, 0, s, *cfg, A); \
        else hipLaunchKernelGGL((KERN<SDM_KERNEL_PARAMETERIZED, false>), GRID, blk, 0, s, *cfg, A); \
        break;                                                                                 \
      case SDM_KERNEL_SIMPLE_GEOMETRIC:                                                               \
        if (brk__) hipLaunchKernelGGL((KERN<SDM_KERNEL_SIMPLE_GEOMETRIC, true>), GRID, blk, 0, s, *cfg, A); \
        else hipLaunchKernelGGL((KERN<SDM_KERNEL_SIMPLE_GEOMETRIC, false>), GRID, blk, 0, s, *cfg, A); \
        break;                                                                                 \
      case SDM_KERNEL_LINEAR:                                                               \
        if (brk__) hipLaunchKernelGGL((KERN<SDM_KERNEL_LINEAR, true>), GRID, blk, 0, s, *cfg, A); \
        else hipLaunchKernelGGL((KERN<SDM_KERNEL_LINEAR, false>), GRID, blk, 0, s, *cfg, A); \
        break;                                                                                 \
      default:                                                                                 \
        if (brk__) hipLaunchKernelGGL((KERN<SDM_KERNEL_CONSTANT, true>), GRID, blk, 0, s, *cfg, A); \
        else hipLaunchKernelGGL((KERN<SDM_KERNEL_CONSTANT, false>), GRID, blk, 0, s, *cfg, A); \
    }                                                                                          \
  } while (0)

// flags: bit 0 = read the control block back at the end; bit 1 = the control block was freshly
// pushed by the host (single-cell bookkeeping has to be initialised)
// fold_counters: false when further steps of the same call follow (sdm_collision_run)
// more_follow: further steps of the same call follow (then the head of the next sub-step is
// launched ahead of each read-back, see `launch_head`)
static int collision_step(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st,
                          sdm_step_result *res, int flags, bool fold_counters, bool more_follow) {
  ARG_TRY(ctx && cfg && st && res);
  ARG_TRY(cfg->n_sd >= 2 && cfg->n_sd < INT32_MAX && cfg->n_cell >= 1 && cfg->n_attr >= 1);
  ARG_TRY(st->idx && st->tmp_idx && st->multiplicity && st->attributes && st->cell_id &&
          st->cell_idx && st->cell_start && st->dt_left && st->stats_dt_min &&
          st->stats_n_substep && st->collision_rate && st->collision_rate_deficit &&
          st->coalescence_rate && st->ctl);
  ARG_TRY(!cfg->enable_breakup || (st->breakup_rate && st->breakup_rate_deficit));
  ARG_TRY((cfg->kernel != SDM_KERNEL_GEOMETRIC && cfg->kernel != SDM_KERNEL_PARAMETERIZED) ||
          (st->gk_a && st->gk_b && cfg->gk_table_len > 0));
  ARG_TRY(cfg->kernel >= SDM_KERNEL_GOLOVIN && cfg->kernel <= SDM_KERNEL_LINEAR);
  ARG_TRY(cfg->adaptive || cfg->substeps >= 1);
  ARG_TRY(cfg->mass_attr >= 0 && cfg->mass_attr < cfg->n_attr);
  ARG_TRY(!(cfg->dt_min <= 0));
  const bool read_back = (flags & 1) != 0;

  const int64_t N = cfg->n_sd, P = N / 2, C = cfg->n_cell;
  // random_generator_optimizer.py:21-25
  int64_t shift = 0;
  if (cfg->optimized_random) {
    const double q = cfg->dt / cfg->dt_min;
    shift = (int64_t)q;
    if ((double)shift < q) shift += 1;
  }
  // an adaptive time step ends within ceil(dt / dt_min) + 1 sub-steps (sdm_hip.h; + 1 for the
  // rounding of the subtractions); the reference's loop has no bound and spins for ever on a state
  // whose cell_start belongs to another permutation - here that is SDM_E_STATE
  ARG_TRY(!cfg->adaptive || (cfg->dt_min == cfg->dt_min && cfg->dt == cfg->dt));
  int64_t max_substeps = INT64_MAX;
  if (cfg->adaptive) {
    const double q = cfg->dt / cfg->dt_min;
    max_substeps = q < 4e18 ? (int64_t)ceil(q) + 2 : INT64_MAX;
    if (ctx->opt_max_substeps > 0 && ctx->opt_max_substeps < max_substeps)
      max_substeps = ctx->opt_max_substeps;  // (tests: SDM_OPT_MAX_SUBSTEPS)
  }
  auto unbounded = [&](int64_t n_done, const int64_t *ctl8) -> int {
    // (the context goes on working: what this call left half-done in it - the single cell's
    // counter slots, which are folded into the counters only at a call's end - is dropped)
    if (ctx->cnt_slots)
      (void)hipMemsetAsync(ctx->cnt_slots, 0, sizeof(int64_t) * SDM_CNT_SLOTS * SDM_CNT_STRIDE,
                           ctx->stream);
    if (ctx->dead_ctr)
      (void)hipMemsetAsync(ctx->dead_ctr, 0, sizeof(unsigned long long) * 32, ctx->stream);
    sdm_set_error("adaptive time step did not end within %lld sub-steps (dt = %g, dt_min = %g): "
                  "the state is inconsistent - a cell_start that does not belong to the "
                  "permutation, or dt_left edited from outside.  Control block {valid %lld, "
                  "work %lld, sorted %lld, healthy %lld, overflow %lld, pairs %lld, max cell "
                  "%lld, events %lld}", (long long)n_done, cfg->dt, cfg->dt_min,
                  (long long)ctl8[0], (long long)ctl8[1], (long long)ctl8[2], (long long)ctl8[3],
                  (long long)ctl8[4], (long long)ctl8[5], (long long)ctl8[6], (long long)ctl8[7]);
    return SDM_E_STATE;
  };
  FusedScratch S = layout(nullptr, cfg);
  int rc = sdm_reserve(ctx, S.total);
  if (rc) return rc;
  S = layout(ctx->arena, cfg);
  rc = sdm_pcg_prepare(ctx, cfg->rng_state_inc);
  if (rc) return rc;
  const u128 rng_state = (((u128)cfg->rng_state_inc[0]) << 64) | cfg->rng_state_inc[1];
  const u128 rng_inc = (((u128)cfg->rng_state_inc[2]) << 64) | cfg->rng_state_inc[3];

  FusedArgs A;
  memset(&A, 0, sizeof(A));
  A.multiplicity = st->multiplicity;
  A.attributes = st->attributes;
  A.cell_id = st->cell_id;
  // (a sharded run with a sharded displacement step keeps every id's own cell apart from the
  // column the permutation is sorted by: sdm_hip.h, sdm_step_state.cell_id_by_id)
  A.cell_id_raw = ctx->cell_id_raw ? ctx->cell_id_raw
                                   : (st->cell_id_by_id ? st->cell_id_by_id : st->cell_id);
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
  A.cell_owned = st->cell_owned;
  A.rng_inc = rng_inc;
  A.rng_tab = ctx->pcg_tab;
  A.rng_aff = ctx->graph_capture ? nullptr : ctx->pcg_aff;
  A.prob = S.prob;
  A.pair_off = S.pair_off;
  A.pair_cid = S.pair_cid;
  A.dt_todo = S.dt_todo;
  A.cell_min = S.cell_min;
  A.list = S.list;
  {
    // two sets of fill counts: a sub-step appends under one, its k_resolve_dense clears the other.
    // Both cleared once per call (the arena is shared with other calls) - or not at all when the
    // previous step of the same run says which set it left clean
    int first_set = 0;
    if (cfg->enable_breakup) {
      if (ctx->lists.active && ctx->lists.owner == (const void *)st && !(flags & 2))
        first_set = ctx->lists.clean_set;
      else
        HIP_TRY(hipMemsetAsync(S.list_count, 0,
                               sizeof(unsigned long long) * 2 * LIST_NL * SDM_CNT_STRIDE,
                               ctx->stream));
    }
    ctx->lists.active = false;
    A.list_count = S.list_count + first_set * LIST_NL * SDM_CNT_STRIDE;
    A.list_count_next = S.list_count + (1 - first_set) * LIST_NL * SDM_CNT_STRIDE;
  }
  A.list_nl = 1;  // the per-cell kernel: one list (a cell's workgroup may hold up to P pairs)
  A.list_cap = P;
  A.block_min = S.block_min;
  A.n_block_min = (int)grid_for((N + 1) / 2);
  A.slots = C == 1 ? ctx->cnt_slots : nullptr;

  uint64_t off = st->rng_offset, off_b = st->rng_offset_breakup;
  uint64_t draw_off = off, draw_off_b = off_b;  // stream positions of the current draw
  int64_t n_sub = 0, n_pairs = 0, swaps = 0;
  int64_t *cur = st->idx, *alt = st->tmp_idx;
  hipStream_t s = ctx->stream;
  const dim3 blk(SDM_BLOCK), one(1);

  A.nm = (double *)st->nm;
  A.nm_wide = mirror_is_wide(cfg);
  if (st->nm && (flags & 2) && !(flags & 4)) {
    hipLaunchKernelGGL(k_nm_init, dim3(grid_for(N)), blk, 0, s, *cfg, A);
    LAUNCH_CHECK();
  }
  if (C == 1 && (flags & 2)) {
    hipLaunchKernelGGL(k_single_cell_init, one, one, 0, s, st->ctl, st->cell_start);
    LAUNCH_CHECK();
  }
  // collision.py:180: dt_left[:] = dt.  One cell: nothing reads dt_left before the first
  // sub-step's k_cells_adaptive, which then does it (one launch less per time step)
  bool fill_pending = cfg->adaptive;  // (multi-cell: decided below, once the route is known)
  int64_t work_host = -1;
  int64_t box_seq = 0;   // sequence number of the control block publication being waited for
  int sorted_host = -1;  // host's knowledge of ctl[CTL_SORTED]
  int64_t max_cell = -1;  // upper bound of the cell sizes during this call (-1: unknown)
  int64_t n_active_cells = C;  // cells this process computes (sharded: those it owns)
  if (flags & 2) st->known_valid = -1;
  int64_t last_ctl[8] = {-1, -1, -1, -1, 0, 0, 0, 0};  // the control block as last read back
  bool have_ctl = false;
  if (C == 1 && st->known_valid >= 0 && cfg->adaptive) {
    // one cell, nothing touched the state since the last read-back: no need to ask the device
    work_host = st->known_valid;
    sorted_host = 1;
  } else if (C > 1 && cfg->adaptive && ctx->carry.active && ctx->carry.owner == (const void *)st &&
             !(flags & 2)) {
    // multi-cell, previous step of the same call: sorted, lengths and cell-size bound known
    work_host = ctx->carry.valid;
    sorted_host = 1;
    max_cell = ctx->carry.max_cell;
    n_active_cells = ctx->carry.n_active_cells;
  } else if (cfg->adaptive || read_back || st->cell_owned) {  // (sharded: the per-cell route)
    bool sorted_now = false;
    if (C > 1 && cfg->croupier_local) {
      // The route depends on the largest cell, which is known only for a sorted state - and a
      // state that comes from the host (after a displacement, say) is not sorted.  So what the
      // first sub-step would do first anyway is done here: dt_left[:] = dt and
      // cell_idx.sort_by_key(dt_left) (collision.py:180-183; adaptive), then the cell_start
      // getter's counting sort, gated on the device by the sorted flag
      if (cfg->adaptive) {
        hipLaunchKernelGGL(k_fill_f64, dim3(grid_for(C)), blk, 0, s, st->dt_left, cfg->dt, C,
                           st->ctl + 6, S.end2 + 5);
        LAUNCH_CHECK();
        fill_pending = false;
        rc = sdm_sort_by_key_async(ctx, st->cell_idx, st->dt_left, C);
        if (rc) return rc;
      } else {
        HIP_TRY(hipMemsetAsync(st->ctl + 6, 0, sizeof(int64_t), s));
        if (st->cell_owned) HIP_TRY(hipMemsetAsync(S.end2 + 5, 0, sizeof(int64_t), s));
      }
      rc = cond_sort(ctx, cfg, st, cur, S, &sorted_host);
      if (rc) return rc;
      sorted_now = true;
      hipLaunchKernelGGL(k_max_cell, dim3(grid_for(C)), blk, 0, s, st->cell_start, C, st->ctl,
                         st->cell_owned, S.end2 + 5);
      LAUNCH_CHECK();
      if (st->cell_owned)
        HIP_TRY(hipMemcpyAsync(ctx->mailbox + 40, S.end2 + 5, sizeof(int64_t),
                               hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(ctx->mailbox + 8, st->ctl, sizeof(int64_t) * 8, hipMemcpyDeviceToHost,
                           s));
    HIP_TRY(hipStreamSynchronize(s));
    if (ctx->mailbox[8 + 7] & 4) {  // (k_max_cell; a bit: other codes may be set beside it)
      HIP_TRY(hipMemsetAsync(st->ctl + 7, 0, sizeof(int64_t), s));
      sdm_set_error("the state was handed over as sorted by cell, but cell_start does not span "
                    "the live super-droplets (a cell_start of another state?): no collision "
                    "computed (dt_left and cell_idx were already reset for the step)");
      return SDM_E_ARG;
    }
    work_host = ctx->mailbox[8 + CTL_WORK];
    sorted_host = (int)ctx->mailbox[8 + CTL_SORTED];
    if (C == 1 && (flags & 2)) sorted_host = 1;
    if (sorted_now && sorted_host == 1) {
      max_cell = ctx->mailbox[8 + 6];
      if (st->cell_owned) n_active_cells = ctx->mailbox[40];
    }
  }
  ctx->carry.active = false;
  const bool cell_path = max_cell >= 0 && max_cell <= CELL_CAP;
  // two workgroups per CU where the kernel for it applies (see k_cell_step2)
  const bool cell2 = cell_path && max_cell <= CELL2_CAP && cfg->n_attr == 1;
  // ... or its 1024-thread shape when there are no more cells to compute than CUs to compute
  // them on (a process of a sharded run with its 128 cells of 1024; a small grid)
  if (ctx->n_cus == 0)
    HIP_TRY(hipDeviceGetAttribute(&ctx->n_cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
  const int shape = ctx->opt_cell_shape;  // (SDM_OPT_CELL_SHAPE: measurements, tests)
  const bool unpacked = cell2 && max_cell > CELL2_CAP / CELL2_PACK;
  const bool cell2w = cell_path && max_cell <= CELL2W_CAP && cfg->n_attr == 1 &&
                      max_cell > CELL2_CAP / CELL2_PACK &&
                      (shape == SDM_CELL_SHAPE_AUTO ? n_active_cells <= ctx->n_cus
                                                    : shape == SDM_CELL_SHAPE_1024);
  // ... or its 256-thread shape, four workgroups per CU, when the cells are small enough for
  // four to fit and there are more than two per CU: +27 % at 1024 cells of 1024, +18 % at 1024 of
  // 2048 (profiles/r04_cell_shapes.json) - the same sixteen wavefronts per CU, in four
  // independent groups whose phases and barriers overlap instead of two
  const bool cell2q = unpacked && !cell2w && max_cell <= CELL2Q_CAP &&
                      (shape == SDM_CELL_SHAPE_AUTO ? n_active_cells > 2 * ctx->n_cus
                                                    : shape == SDM_CELL_SHAPE_256);
  if (fill_pending && C > 1 && !cell_path) {  // the per-cell route's k_cells_turn does it itself
    hipLaunchKernelGGL(k_fill_f64, dim3(grid_for(C)), blk, 0, s, st->dt_left, cfg->dt, C);
    LAUNCH_CHECK();
    fill_pending = false;
  }
  if (cell_path) {
    // > 64 KiB of dynamic LDS must be opted into, per kernel and per device (one ctx = one device)
    if (!ctx->cell_attr_done) {
      const int lds = CELL_LDS_BYTES;
#define CELL_ATTR(K, B) HIP_TRY(hipFuncSetAttribute((const void *)k_cell_step<K, B>, \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds))
      CELL_ATTR(SDM_KERNEL_GOLOVIN, false); CELL_ATTR(SDM_KERNEL_GOLOVIN, true);
      CELL_ATTR(SDM_KERNEL_GEOMETRIC, false); CELL_ATTR(SDM_KERNEL_GEOMETRIC, true);
      CELL_ATTR(SDM_KERNEL_CONSTANT, false); CELL_ATTR(SDM_KERNEL_CONSTANT, true);
      CELL_ATTR(SDM_KERNEL_PARAMETERIZED, false); CELL_ATTR(SDM_KERNEL_PARAMETERIZED, true);
      CELL_ATTR(SDM_KERNEL_SIMPLE_GEOMETRIC, false); CELL_ATTR(SDM_KERNEL_SIMPLE_GEOMETRIC, true);
      CELL_ATTR(SDM_KERNEL_LINEAR, false); CELL_ATTR(SDM_KERNEL_LINEAR, true);
#undef CELL_ATTR
#define CELL2_ATTR(K)                                                                           \
  HIP_TRY(hipFuncSetAttribute((const void *)k_cell_step2<K, false, 1, CELL2_THREADS>,           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, CELL2_LDS_BYTES));    \
  HIP_TRY(hipFuncSetAttribute((const void *)k_cell_step2<K, true, 1, CELL2_THREADS>,            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, CELL2_LDS_BYTES));    \
  HIP_TRY(hipFuncSetAttribute((const void *)k_cell_step2<K, false, CELL2_PACK, CELL2_THREADS>,  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, CELL2_LDS_BYTES));    \
  HIP_TRY(hipFuncSetAttribute((const void *)k_cell_step2<K, true, CELL2_PACK, CELL2_THREADS>,   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, CELL2_LDS_BYTES));    \
  HIP_TRY(hipFuncSetAttribute((const void *)k_cell_step2<K, false, 1, CELL2W_THREADS>,          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, CELL2W_LDS_BYTES));   \
  HIP_TRY(hipFuncSetAttribute((const void *)k_cell_step2<K, true, 1, CELL2W_THREADS>,           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, CELL2W_LDS_BYTES))
      CELL2_ATTR(SDM_KERNEL_GOLOVIN); CELL2_ATTR(SDM_KERNEL_GEOMETRIC);
      CELL2_ATTR(SDM_KERNEL_CONSTANT); CELL2_ATTR(SDM_KERNEL_PARAMETERIZED);
      CELL2_ATTR(SDM_KERNEL_SIMPLE_GEOMETRIC); CELL2_ATTR(SDM_KERNEL_LINEAR);
#undef CELL2_ATTR
      ctx->cell_attr_done = true;
    }
  }
  // One adaptive cell, plain random numbers: what a sub-step does before its first global
  // dependency - draw, shuffle build, permutation + probabilities (k_pair_prob) - is the same
  // whether it continues this time step or opens the next one (the generator stream simply goes
  // on; dt_left enters only afterwards, in k_cells_adaptive).  So when more steps follow, that
  // head is launched right after the compaction of the current sub-step, *before* the host waits
  // for the control block: the GPU works through the ~12 us of the read-back instead of idling.
  // The head touches nothing but scratch and the spare permutation buffer, so it can be dropped
  // (stream positions restored) in the one case it was not needed: no super-droplet left.
  const bool head_ok = C == 1 && cfg->adaptive && !cfg->optimized_random && cfg->croupier_local &&
                       sdm_shuffle_can_split(N, false) && N >= 4096;
  bool head_done = false;
  uint64_t head_off_before = off, head_off_b_before = off_b;
  auto launch_head = [&]() -> int {
    head_off_before = off;
    head_off_b_before = off_b;
    draw_off = off;
    draw_off_b = off_b;
    off += (uint64_t)(N + shift + P);
    if (cfg->enable_breakup) off_b += (uint64_t)P;
    A.s_rand = sdm_pcg_advance_host(rng_state, rng_inc, draw_off + (uint64_t)(N + shift));
    A.s_rand_b = sdm_pcg_advance_host(rng_state, rng_inc, draw_off_b);
    ShuffleViews views;
    const int r = sdm_shuffle_build_async(ctx, S.shuffle, cur, st->cell_start, C,
                                          st->cell_start + C, N, cfg->rng_state_inc, draw_off,
                                          &views, N);
    if (r) return r;
    A.rec = views.rec;
    A.rec_fmt = views.fmt;
    A.ovf_head = views.ovf_head;
    A.ovf_next = views.ovf_next;
    A.idx_prev = cur;
    { int64_t *t = cur; cur = alt; alt = t; }
    ++swaps;
    A.idx = cur;
    PhaseScope ph(ctx, SDM_PHASE_PAIR_PROB);
    DISPATCH_PAIR(k_pair_prob, dim3(grid_for((N + 1) / 2)));
    LAUNCH_CHECK();
    return SDM_OK;
  };
  if (head_ok && cfg->enable_breakup) {
    A.list_nl = LIST_NL;
    A.list_cap = S.flat_list_cap;
  }
  if (ctx->ahead.active) {  // left by the previous step of this call
    ctx->ahead.active = false;
    if (head_ok && work_host != 0 && ctx->ahead.owner == (const void *)st &&
        ctx->ahead.cur == st->tmp_idx && ctx->ahead.alt == st->idx) {
      A.s_rand = ctx->ahead.s_rand;
      A.s_rand_b = ctx->ahead.s_rand_b;
      A.rec = ctx->ahead.rec;
      A.rec_fmt = ctx->ahead.rec_fmt;
      A.ovf_head = (const int32_t *)ctx->ahead.ovf_head;
      A.ovf_next = (const int32_t *)ctx->ahead.ovf_next;
      cur = ctx->ahead.cur;
      alt = ctx->ahead.alt;
      A.idx_prev = alt;
      A.idx = cur;
      swaps = 1;
      head_done = true;
    } else {  // not needed after all: its draw goes back to the streams
      off = ctx->ahead.off_before;
      off_b = ctx->ahead.off_b_before;
    }
  }
  auto launch_cell_kernel = [&](const CellArgs &X) -> int {
    PhaseScope ph(ctx, SDM_PHASE_PAIR_UPDATE);
    const bool brk = cfg->enable_breakup != 0;
    // small cells: CELL2_PACK of them per workgroup (k_cell_step2)
    const bool packed = cell2 && max_cell >= 0 && max_cell <= CELL2_CAP / CELL2_PACK;
    const bool tiny = packed && max_cell <= CELL2T_CAP / CELL2_PACK &&
                      ctx->opt_cell_shape != SDM_CELL_SHAPE_512;  // (forced 512: the 704 cap)
    const dim3 grid((unsigned)(C + X.n_tail_blocks));
    const dim3 grid_p((unsigned)((C + CELL2_PACK - 1) / CELL2_PACK + X.n_tail_blocks));
#define CELL_LAUNCH(K)                                                                        \
  do {                                                                                        \
    if (tiny && brk) hipLaunchKernelGGL((k_cell_step2<K, true, CELL2_PACK, CELL2_THREADS, true>), grid_p, dim3(CELL2_THREADS), CELL2T_LDS_BYTES, s, *cfg, A, X); \
    else if (tiny) hipLaunchKernelGGL((k_cell_step2<K, false, CELL2_PACK, CELL2_THREADS, true>), grid_p, dim3(CELL2_THREADS), CELL2T_LDS_BYTES, s, *cfg, A, X); \
    else if (packed && brk) hipLaunchKernelGGL((k_cell_step2<K, true, CELL2_PACK, CELL2_THREADS>), grid_p, dim3(CELL2_THREADS), CELL2_LDS_BYTES, s, *cfg, A, X); \
    else if (packed) hipLaunchKernelGGL((k_cell_step2<K, false, CELL2_PACK, CELL2_THREADS>), grid_p, dim3(CELL2_THREADS), CELL2_LDS_BYTES, s, *cfg, A, X); \
    else if (cell2w && brk) hipLaunchKernelGGL((k_cell_step2<K, true, 1, CELL2W_THREADS>), grid, dim3(CELL2W_THREADS), CELL2W_LDS_BYTES, s, *cfg, A, X); \
    else if (cell2w) hipLaunchKernelGGL((k_cell_step2<K, false, 1, CELL2W_THREADS>), grid, dim3(CELL2W_THREADS), CELL2W_LDS_BYTES, s, *cfg, A, X); \
    else if (cell2q && brk) hipLaunchKernelGGL((k_cell_step2<K, true, 1, CELL2Q_THREADS>), grid, dim3(CELL2Q_THREADS), CELL2Q_LDS_BYTES, s, *cfg, A, X); \
    else if (cell2q) hipLaunchKernelGGL((k_cell_step2<K, false, 1, CELL2Q_THREADS>), grid, dim3(CELL2Q_THREADS), CELL2Q_LDS_BYTES, s, *cfg, A, X); \
    else if (cell2 && brk) hipLaunchKernelGGL((k_cell_step2<K, true, 1, CELL2_THREADS>), grid, dim3(CELL2_THREADS), CELL2_LDS_BYTES, s, *cfg, A, X); \
    else if (cell2) hipLaunchKernelGGL((k_cell_step2<K, false, 1, CELL2_THREADS>), grid, dim3(CELL2_THREADS), CELL2_LDS_BYTES, s, *cfg, A, X); \
    else if (brk) hipLaunchKernelGGL((k_cell_step<K, true>), grid, dim3(CELL_THREADS), CELL_LDS_BYTES, s, *cfg, A, X); \
    else hipLaunchKernelGGL((k_cell_step<K, false>), grid, dim3(CELL_THREADS), CELL_LDS_BYTES, s, *cfg, A, X);    \
  } while (0)
    switch (cfg->kernel) {
      case SDM_KERNEL_GOLOVIN: CELL_LAUNCH(SDM_KERNEL_GOLOVIN); break;
      case SDM_KERNEL_GEOMETRIC: CELL_LAUNCH(SDM_KERNEL_GEOMETRIC); break;
      case SDM_KERNEL_PARAMETERIZED: CELL_LAUNCH(SDM_KERNEL_PARAMETERIZED); break;
      case SDM_KERNEL_SIMPLE_GEOMETRIC: CELL_LAUNCH(SDM_KERNEL_SIMPLE_GEOMETRIC); break;
      case SDM_KERNEL_LINEAR: CELL_LAUNCH(SDM_KERNEL_LINEAR); break;
      default: CELL_LAUNCH(SDM_KERNEL_CONSTANT);
    }
#undef CELL_LAUNCH
    LAUNCH_CHECK();
    return SDM_OK;
  };
  // ---- sharded mode (sdm_hip.h): this process computes the cells it owns on arrays of the global
  // shape.  After the kernels of a sub-step: sum over the processes of the owned cells' dt_left
  // and of "a super-droplet died"; if one died anywhere, the permutation is put together again
  // from the processes' segments, so that the compaction and the counting sort that follow run
  // on identical, complete data everywhere.
  const bool sharded = st->cell_owned != nullptr;
  if (sharded) {
    if ((!st->exchange && !ctx->comm) || !st->xchg_cells || !st->xchg_idx) {
      sdm_set_error("sharded mode: an exchange callback (or a communicator: sdm_comm_init) and "
                    "the two exchange buffers are required");
      return SDM_E_ARG;
    }
    if (st->shard_world < 1 || st->shard_world > 64 || st->shard_rank < 0 ||
        st->shard_rank >= st->shard_world) {
      sdm_set_error("sharded mode: shard_rank / shard_world out of range");
      return SDM_E_ARG;
    }
    // (cells beyond the per-cell kernels' capacity and the global croupier take the generic
    // kernels, as in a one-process run: those skip the pairs of cells this process does not own,
    // the per-sub-step exchange - owned cells' dt_left, deaths - is the one of sdm_hip.h.  The
    // invariant holds under the global croupier as well: the shuffle permutes POSITIONS, a
    // process's own super-droplets stand at their true positions, and the stable counting sort
    // orders a cell's members by where the shuffle put them)
  }
  // dead positions of this process (scratch: the sort's output buffer is free between sorts)
  int64_t *shard_dead_pos = S.sorted_buf;
  unsigned long long *shard_n_dead = (unsigned long long *)(S.end2 + 4);
  const int world = sharded ? st->shard_world : 1, my_rank = sharded ? st->shard_rank : 0;
  unsigned long long *shard_n_listed = (unsigned long long *)(S.end2 + 6);
  A.n_dead = sharded ? shard_n_dead : nullptr;
  // owned cells' dt_left + deaths (counted by the kernels that flagged them), summed
  auto shard_cells = [&]() -> int {
    const dim3 g((unsigned)grid_for(C + 1 + world));
    hipLaunchKernelGGL(k_shard_pack, g, blk, 0, s, A, C, cfg->adaptive, st->xchg_cells,
                       shard_n_dead, my_rank, world);
    LAUNCH_CHECK();
    {
      const int r = sdm_exchange(ctx, st->exchange, st->exchange_user, SDM_XCHG_SUM_F64,
                                 st->xchg_cells, C + 1 + world);
      if (r) return r;
    }
    hipLaunchKernelGGL(k_shard_unpack, g, blk, 0, s, A, C, cfg->adaptive, st->xchg_cells);
    LAUNCH_CHECK();
    return SDM_OK;
  };
  // A super-droplet died somewhere: every process learns WHERE (positions are global) - one
  // all-gather of the dead positions, as a sum of rank-disjoint slices of exactly their total
  // number - and flags those positions in its own permutation; the compaction that follows then
  // does the reference's swap-from-the-end (collisions_methods.py:664-680) on every process alike.
  // (Round 2 summed the whole masked permutation here: n_sd int64 per death.)
  auto shard_dead = [&](int64_t *perm) -> int {
    // (straight into this frame: the pinned mailbox has 16 words before the polled control block,
    // 1 + world doubles need up to 65)
    double counts[1 + 64];
    HIP_TRY(hipMemcpyAsync(counts, st->xchg_cells + C, sizeof(double) * (size_t)(1 + world),
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int64_t total = (int64_t)counts[0];
    if (total <= 0) return SDM_OK;
    int64_t before = 0;
    for (int r = 0; r < my_rank; ++r) before += (int64_t)counts[1 + r];
    const int64_t mine = (int64_t)counts[1 + my_rank];
    if (total > N || before + mine > total) {
      sdm_set_error("sharded mode: inconsistent death counts (%lld of %lld)", (long long)mine,
                    (long long)total);
      return SDM_E_HIP;
    }
    HIP_TRY(hipMemsetAsync(st->xchg_idx, 0, sizeof(int64_t) * (size_t)total, s));
    if (mine > 0) {
      // the positions of this process's dead: flagged entries of the segments it owns
      HIP_TRY(hipMemsetAsync(shard_n_listed, 0, sizeof(unsigned long long), s));
      hipLaunchKernelGGL(k_shard_dead_list, dim3((unsigned)C), blk, 0, s, A, (const int64_t *)perm,
                         C, N, shard_dead_pos, shard_n_listed);
      LAUNCH_CHECK();
      hipLaunchKernelGGL(k_shard_dead_place, dim3(grid_for(mine)), blk, 0, s, shard_dead_pos, mine,
                         st->xchg_idx + before);
      LAUNCH_CHECK();
    }
    {
      const int r = sdm_exchange(ctx, st->exchange, st->exchange_user, SDM_XCHG_SUM_I64,
                                 st->xchg_idx, total);
      if (r) return r;
    }
    hipLaunchKernelGGL(k_shard_flag, dim3(grid_for(total)), blk, 0, s, perm, st->xchg_idx, total,
                       N);
    LAUNCH_CHECK();
    return SDM_OK;
  };
  auto shard_sync = [&](int64_t *perm) -> int {  // both
    const int r = shard_cells();
    return r ? r : shard_dead(perm);
  };
  // Multi-cell per-cell route, adaptive.  A sub-step is two launches: k_cells_turn (ends the
  // previous sub-step: bookkeeping, working length, publication of its control block; opens this
  // one: cell order, per-cell init, the decision whether it runs at all) and the cell kernel; in
  // sharded runs an all-reduce MIN of the cell minima (and minus the deaths per segment) follows.
  // Sub-step k + 1 is launched before the host waits for the control block of sub-step k, which
  // k + 1's own first kernel publishes; if k left nothing to do, un-sorted the state or killed a
  // super-droplet, the whole of k + 1 falls through on the device (gate word) and the host takes
  // back what it had booked for it (stream positions, buffer exchange).  A death is dealt with
  // here: (sharded: exchange of the dead positions,) compaction, re-sort, a fresh working length.
  if (cell_path && cfg->adaptive && work_host != 0) {
    int64_t launched = 0;  // sub-steps launched in this time step (the draw window shifts by it)
    int64_t turn = 0;      // turn kernels launched in this call of the step
    bool shard_resorted = false;
    // ping-pong (k_cells_turn): dt_left between the caller's array and scratch, the cell minima
    // (+ deaths per segment) between the two halves of the exchange buffer / of the scratch
    double *left_buf[2] = {st->dt_left, S.dt_todo};
    double *min_base = sharded ? st->xchg_cells : S.cell_min;
    double *min_buf[2] = {min_base, min_base + 2 * C};
    int left_cur = 0;  // left_buf[left_cur] holds the current dt_left
    int64_t *gate_words = S.end2 + 2;  // two words (turn parity)
    int64_t pending_seq = 0;           // publication number of the sub-step not yet ended
    double *last_min = nullptr;        // the buffer the sub-step not yet ended reduced into
    bool segments_changed = true;      // k_cells_turn writes FusedArgs::seg_cid / seg_owned anew
    auto launch_turn = [&](bool gated, bool end_only) -> int {
      PhaseScope ph(ctx, SDM_PHASE_CELLS_PRE);
      TurnArgs T;
      memset(&T, 0, sizeof(T));
      T.left_in = left_buf[left_cur];
      T.left_out = left_buf[left_cur ^ 1];
      T.min_in = min_buf[(turn + 1) & 1];
      T.min_out = min_buf[turn & 1];
      T.cell_idx = st->cell_idx;
      T.gate = gate_words;
      T.turn = turn;
      T.first = turn == 0;
      T.fresh = fill_pending ? 1 : 0;
      T.gated = gated ? 1 : 0;
      T.end_only = end_only ? 1 : 0;
      T.sharded = sharded ? 1 : 0;
      T.box = ctx->box_dev;
      T.seq = pending_seq;
      if (segments_changed) {  // (which cell a segment holds changes with a sort only)
        T.perm = cur;
        T.seg_cid = S.seg_cid;
        T.seg_owned = sharded ? S.seg_owned : nullptr;
        segments_changed = false;
      }
      if (C <= 2048)
        hipLaunchKernelGGL(k_cells_turn<1>, dim3((unsigned)C), blk, 0, s, *cfg, A, T);
      else if (C <= 8192)
        hipLaunchKernelGGL(k_cells_turn<8>, dim3((unsigned)((C + 7) / 8)), blk, 0, s, *cfg, A, T);
      else
        hipLaunchKernelGGL(k_cells_turn<16>, dim3((unsigned)((C + 15) / 16)), blk, 0, s, *cfg, A,
                           T);
      LAUNCH_CHECK();
      fill_pending = false;
      left_cur ^= 1;
      A.dt_left = left_buf[left_cur];
      A.cell_min = T.min_out;
      A.seg_deaths = sharded ? T.min_out + C : nullptr;
      A.seg_owned = sharded ? S.seg_owned : nullptr;
      A.seg_cid = S.seg_cid;
      ++turn;
      return SDM_OK;
    };
    // returns the publication number under which the sub-step's control block will appear
    auto launch_substep = [&](bool gated, int64_t *seq_out) -> int {
      if (!cfg->optimized_random || launched == 0) {  // (c), as in the loop below
        draw_off = off;
        draw_off_b = off_b;
        off += (uint64_t)(N + shift + P);
        if (cfg->enable_breakup) off_b += (uint64_t)P;
      }
      A.s_rand = sdm_pcg_advance_host(rng_state, rng_inc, draw_off + (uint64_t)(N + shift));
      A.s_rand_b = sdm_pcg_advance_host(rng_state, rng_inc, draw_off_b);
      const uint64_t u01_off = draw_off + (uint64_t)(cfg->optimized_random ? launched : 0);
      if (launched == 0 || shard_resorted) segments_changed = true;
      int r = launch_turn(gated, false);  // (publishes the control block of the sub-step before)
      if (r) return r;
      CellArgs X;
      memset(&X, 0, sizeof(X));
      X.idx_in = cur;
      X.idx_out = alt;
      X.s_u01 = sdm_pcg_advance_host(rng_state, rng_inc, u01_off);
      X.n_tail_blocks = 64;
      X.gate = gate_words + ((turn - 1) & 1);
      X.copy_others = sharded && launched == 0;  // (see CellArgs; once per time step suffices...
      if (sharded && shard_resorted) { X.copy_others = 1; shard_resorted = false; }  // ...or sort)
      A.idx = alt;
      r = launch_cell_kernel(X);
      if (r) return r;
      { int64_t *t = cur; cur = alt; alt = t; }
      ++swaps;
      if (cfg->enable_breakup) {
        PhaseScope ph(ctx, SDM_PHASE_PAIR_UPDATE);
        const int64_t chunks = (A.list_cap + SDM_BLOCK - 1) / SDM_BLOCK;
        hipLaunchKernelGGL(k_resolve_dense, dim3((unsigned)(A.list_nl * chunks)), blk, 0, s, *cfg,
                           A);
        LAUNCH_CHECK();
        std::swap(A.list_count, A.list_count_next);
      }
      last_min = A.cell_min;
      if (sharded) {
        // every process's cell minima and deaths per segment in one reduction (a sub-step that
        // fell through reduces stale words nobody reads: the number of collectives of a step
        // does not depend on what the device decided)
        r = sdm_exchange(ctx, st->exchange, st->exchange_user, SDM_XCHG_MIN_F64, A.cell_min, 2 * C);
        if (r) return r;
      }
      *seq_out = pending_seq = ++ctx->poll_seq;
      ++launched;
      return SDM_OK;
    };
    // A super-droplet died in sub-step k (sharded: somewhere): the positions of all the dead reach
    // every process - one all-gather, as a sum of segment-ordered disjoint slices of exactly their
    // number - and are flagged in its permutation (in a segment of another process that removes
    // SOME member of the cell: the invariant of sdm_hip.h); the compaction follows
    int64_t listed_dead = 0;  // (dead positions standing in st->xchg_idx after shard_dead_by_segment)
    auto shard_dead_by_segment = [&](int64_t *perm, const double *mins) -> int {
      listed_dead = 0;
      int64_t *offsets = S.seg_src, *total_dev = S.end2 + 6;
      hipLaunchKernelGGL(k_shard_dead_offsets, one, dim3(1024), 0, s, mins + C, C, offsets,
                         total_dev);
      LAUNCH_CHECK();
      HIP_TRY(hipMemcpyAsync(ctx->mailbox, total_dev, sizeof(int64_t), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      const int64_t total = ctx->mailbox[0];
      if (total <= 0) return SDM_OK;
      if (total > N) {
        sdm_set_error("sharded mode: inconsistent death counts (%lld)", (long long)total);
        return SDM_E_HIP;
      }
      HIP_TRY(hipMemsetAsync(st->xchg_idx, 0, sizeof(int64_t) * (size_t)total, s));
      hipLaunchKernelGGL(k_shard_dead_list_by_segment, dim3((unsigned)C), blk, 0, s,
                         (const int64_t *)st->cell_start, mins + C, (const int64_t *)offsets,
                         (const int64_t *)perm, N, st->xchg_idx);
      LAUNCH_CHECK();
      const int r = sdm_exchange(ctx, st->exchange, st->exchange_user, SDM_XCHG_SUM_I64,
                                 st->xchg_idx, total);
      if (r) return r;
      hipLaunchKernelGGL(k_shard_flag, dim3(grid_for(total)), blk, 0, s, perm, st->xchg_idx, total,
                         N);
      LAUNCH_CHECK();
      listed_dead = total;
      return SDM_OK;
    };
    int64_t seq_k = 0;
    rc = launch_substep(false, &seq_k);
    if (rc) return rc;
    for (;;) {
      const uint64_t keep[4] = {off, off_b, draw_off, draw_off_b};
      int64_t seq_next = 0;
      double *min_k = last_min;  // what sub-step k reduced into
      // (not in timing mode: a sub-step that falls through would count as a launch; there the
      // sub-step is ended by a turn kernel of its own.  Sharded runs always launch ahead: the
      // number of collectives of a step must not depend on a switch one process may have set
      // and another not)
      const bool ahead = !ctx->timing || sharded;
      if (ahead) {
        rc = launch_substep(true, &seq_next);
      } else {
        rc = launch_turn(false, true);
      }
      if (rc) return rc;
      bool taken_back = !ahead;
      auto take_back = [&]() {  // the sub-step launched ahead fell through on the device
        if (taken_back) return;
        taken_back = true;
        ++ctx->stats[SDM_STAT_SUBSTEPS_TAKEN_BACK];
        off = keep[0]; off_b = keep[1]; draw_off = keep[2]; draw_off_b = keep[3];
        { int64_t *t = cur; cur = alt; alt = t; }
        --swaps;
        --launched;
      };
      {
        PhaseScope ph(ctx, SDM_PHASE_ADAPTIVE_END);
        rc = sdm_read_box(ctx, seq_k, last_ctl);
        if (rc) return rc;
      }
      pending_seq = ahead ? seq_next : 0;
      have_ctl = true;
      ++n_sub;
      n_pairs += work_host / 2;
      if (last_ctl[7] & SDM_CTL7_ERROR_MASK) {  // device-side failure: the caller raises
        take_back();
        break;
      }
      if (n_sub > max_substeps) {
        // (what was launched ahead drains by itself: every kernel of a sub-step ends; the stream
        // positions it booked are taken back so that the state's offsets stay those of n_sub)
        take_back();
        HIP_TRY(hipStreamSynchronize(s));
        return unbounded(n_sub, last_ctl);
      }
      if (last_ctl[CTL_HEALTHY] == 0) {
        // a super-droplet died (the sub-step launched ahead fell through: k_cells_turn's gate).
        // Sharded: every process flags the dead positions in its own permutation first
        take_back();
        pending_seq = 0;  // (the fallen-through sub-step has nothing to publish)
        if (sharded) {
          rc = shard_dead_by_segment(cur, min_k);
          if (rc) return rc;
        }
        {
          PhaseScope ph(ctx, SDM_PHASE_SANITIZE);
          // (sharded: the positions to remove are known as a list - no pass over the permutation)
          if (sharded && sdm_compact_listed_fits(listed_dead))
            rc = sdm_compact_listed_async(ctx, S.compact, cur, st->xchg_idx, listed_dead, N, N,
                                          st->ctl, S.cctl, nullptr);
          else
            rc = sdm_compact_fused_async(ctx, S.compact, st->multiplicity, cur, N, N, st->ctl,
                                         S.cctl, nullptr, true);
          if (rc) return rc;
        }
        // sort by cell, then the end of the working range from the new cell_start
        // (particle_attributes.py cell_start getter)
        sorted_host = 0;
        shard_resorted = true;
        hipLaunchKernelGGL(k_reset_work, one, one, 0, s, st->ctl);  // sanitize left work = valid
        LAUNCH_CHECK();
        {
          // a handful of deaths in a sorted state: the re-sort has a closed form (index.hip;
          // SDM_OPT_RESORT says whether to ask for it)
          bool closed_form = false;
          PhaseScope ph(ctx, SDM_PHASE_SORT);
          // (asking costs a round trip: after a refusal - too many deaths at once, a tail that
          // spans cells - the next few compactions of this call go straight to the counting sort)
          if (ctx->opt_resort == SDM_RESORT_COUNTING_SORT) {
          } else if (ctx->opt_resort == SDM_RESORT_AUTO && ctx->resort_backoff > 0) {
            --ctx->resort_backoff;
            ++ctx->stats[SDM_STAT_RESORT_SKIPPED];
          } else {
            rc = sdm_resort_plan(ctx, S.compact, N, S.cctl, st->ctl, st->cell_start, C,
                                 S.resort_plan, &closed_form);
            if (rc) return rc;
            if (!closed_form) {
              ctx->resort_backoff = 16;
              ++ctx->stats[SDM_STAT_RESORT_REFUSED];
            }
          }
          if (closed_form) {
            rc = sdm_resort_after_compaction_async(ctx, S.compact, N, st->ctl, cur, S.sorted_buf,
                                                   st->cell_start, S.cs_tmp, st->cell_id,
                                                   st->cell_idx, C, S.resort_plan, S.seg_size,
                                                   S.seg_src);
            if (rc) return rc;
            sorted_host = 1;
            ++ctx->stats[SDM_STAT_RESORT_CLOSED_FORM];
          } else {
            ++ctx->stats[SDM_STAT_RESORT_COUNTING_SORT];
          }
        }
        rc = cond_sort(ctx, cfg, st, cur, S, &sorted_host);  // (no-op when sorted_host == 1)
        if (rc) return rc;
        const int64_t seq = ++ctx->poll_seq;
        rc = sdm_adaptive_end_async(ctx, left_buf[left_cur], C, st->cell_start, S.end2,
                                    S.end2 + 1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_set_work, one, one, 0, s, st->ctl, S.end2 + 1, ctx->box_dev, seq);
        LAUNCH_CHECK();
        rc = sdm_read_box(ctx, seq, last_ctl);
        if (rc) return rc;
        work_host = last_ctl[CTL_WORK];
        if (work_host == 0) break;
        rc = launch_substep(false, &seq_k);
        if (rc) return rc;
        continue;
      }
      sorted_host = 1;
      work_host = last_ctl[CTL_WORK];
      if (work_host == 0) {
        take_back();
        break;
      }
      if (ahead) {
        seq_k = seq_next;
      } else {
        rc = launch_substep(false, &seq_k);
        if (rc) return rc;
      }
    }
    A.dt_left = st->dt_left;
    A.seg_deaths = nullptr;
    A.seg_owned = nullptr;
    A.seg_cid = nullptr;
    if (left_cur != 0)  // the step's last dt_left stands in the scratch half of the ping-pong
      HIP_TRY(hipMemcpyAsync(st->dt_left, left_buf[1], sizeof(double) * (size_t)C,
                             hipMemcpyDeviceToDevice, s));
  }
  // one cell whose shuffle is left to the pair kernels (records only, see (d))
  const bool split_one = C == 1 && cfg->croupier_local && sdm_shuffle_can_split(N, false);
  for (;;) {
    if (!cfg->adaptive && n_sub >= cfg->substeps) break;
    if (cfg->adaptive && work_host == 0) break;
    bool sort_ahead = false;  // this sub-step's pair kernel sorts the next one's events
    if (sharded)  // (the per-cell adaptive route: k_cells_turn does it)
      HIP_TRY(hipMemsetAsync(shard_n_dead, 0, sizeof(unsigned long long), s));
    if (head_ok) {
      if (!head_done) {
        rc = launch_head();
        if (rc) return rc;
      }
      head_done = false;
    } else {
    // (a) collision.py:183 cell_idx.sort_by_key(dt_left)
    if (cfg->adaptive && C > 1 && !cell_path) {  // (per-cell route: k_cells_turn, above)
      rc = sdm_sort_by_key_async(ctx, st->cell_idx, st->dt_left, C);
      if (rc) return rc;
    }
    // (b) cell_start getter: counting sort if unsorted (particle_attributes.py:51-55) - asked
    // for by the LOCAL shuffle only (`permutation`, :98-105: `idx.shuffle(u01, parts=cell_start)`);
    // the global shuffle permutes the state as it stands, sorted or not, and the sort follows it
    // (found by tests/fuzz_parity.py: an unsorted state from the host, several cells, global
    // croupier - no golden had covered the combination)
    if (cfg->croupier_local) {
      rc = cond_sort(ctx, cfg, st, cur, S, &sorted_host);
      if (rc) return rc;
    }
    // (c) random numbers (random_generator_optimizer.py:37-48): a draw = pairs_rand (N + shift)
    // then rand (P) from the collision generator, P from each breakup generator
    if (!cfg->optimized_random || n_sub == 0) {
      draw_off = off;
      draw_off_b = off_b;
      off += (uint64_t)(N + shift + P);
      if (cfg->enable_breakup) off_b += (uint64_t)P;
    }
    A.s_rand = sdm_pcg_advance_host(rng_state, rng_inc, draw_off + (uint64_t)(N + shift));
    A.s_rand_b = sdm_pcg_advance_host(rng_state, rng_inc, draw_off_b);
    if (ctx->graph_capture) {  // stream positions from the device (see common.h: gwords)
      A.s_rand = rng_state;
      A.s_rand_b = rng_state;
      A.dev_off = ctx->gwords;
      A.rand_extra = (uint64_t)(N + shift);
    }
    // (d) permutation (particle_attributes.py:98-105).  shuffle_local visits every cell of
    // cell_start, also those beyond a cut working length (index_methods.py:35); shuffle_global
    // covers the working length (index.py:43-45).  u01[i] = draw (window shift + i).
    const int64_t *p_shuffle_len = cfg->croupier_local ? st->cell_start + C : st->ctl + CTL_WORK;
    const uint64_t u01_off = draw_off + (uint64_t)(cfg->optimized_random ? n_sub : 0);
    const bool split = C == 1 && cfg->croupier_local && sdm_shuffle_can_split(N, false);
    if (cell_path) {
      // one workgroup per cell does the whole sub-step of its cell (see k_cell_step); adaptive
      // steps took the loop above
      CellArgs X;
      memset(&X, 0, sizeof(X));
      X.idx_in = cur;
      X.idx_out = alt;
      X.s_u01 = sdm_pcg_advance_host(rng_state, rng_inc, u01_off);
      X.n_tail_blocks = 64;
      X.gate = nullptr;
      X.copy_others = sharded ? 1 : 0;
      A.idx = alt;
      {
        const int r = launch_cell_kernel(X);
        if (r) return r;
      }
      { int64_t *t = cur; cur = alt; alt = t; }
      ++swaps;
    } else if (split) {
      // single cell: event records only; the pair kernels walk them (2 positions per thread)
      ShuffleViews views;
      // the pair kernel of the previous sub-step of this run sorted this one's events too
      // (k_pair_all_sort): no k_bin_sort, and the compaction it skipped is the build's to do
      SortPrologue prologue;
      const bool presorted = ctx->presorted.active && ctx->presorted.owner == (const void *)st;
      ctx->presorted.active = false;
      if (presorted)
        sdm_compact_as_prologue(ctx, S.compact, st->multiplicity, cur, N, N, st->ctl, S.cctl,
                                st->cell_start, &prologue);
      rc = sdm_shuffle_build_async(ctx, S.shuffle, cur, st->cell_start, C, p_shuffle_len, N,
                                   cfg->rng_state_inc, u01_off, &views, N,
                                   ctx->graph_capture ? ctx->gwords : nullptr,
                                   presorted ? &prologue : nullptr);
      if (rc) return rc;
      A.rec = views.rec;
      A.rec_fmt = views.fmt;
      A.ovf_head = views.ovf_head;
      A.ovf_next = views.ovf_next;
      A.idx_prev = cur;
    } else {
      rc = sdm_shuffle_async(ctx, S.shuffle, alt, cur, nullptr, st->cell_start, C, p_shuffle_len,
                             N, !cfg->croupier_local, N, cfg->rng_state_inc, u01_off);
      if (rc) return rc;
    }
    if (!cell_path) {
      int64_t *t = cur; cur = alt; alt = t;
      ++swaps;
    }
    if (!cfg->croupier_local && C > 1) {
      hipLaunchKernelGGL(k_mark_unsorted, one, one, 0, s, st->ctl);
      LAUNCH_CHECK();
      sorted_host = 0;
      rc = cond_sort(ctx, cfg, st, cur, S, &sorted_host);
      if (rc) return rc;
    }
    A.idx = cur;
    if (cfg->enable_breakup && !cell_path) {
      A.list_nl = LIST_NL;
      A.list_cap = S.flat_list_cap;
    }
    }  // !head_ok
    // one adaptive cell: the kernels below list the positions of the dead for the compaction
    const bool list_dead = C == 1 && cfg->adaptive && !sharded && !ctx->graph_capture;
    if (list_dead) {
      A.dead_pos = ctx->dead_pos;
      A.dead_count = ctx->dead_ctr + (ctx->dead_seq & 1) * 16;
    }
    // (e)+(f) probabilities, gamma, update
    if (cell_path) {
      // done by k_cell_step above
    } else if (!cfg->adaptive) {
      PhaseScope ph(ctx, SDM_PHASE_PAIR_UPDATE);
      // another sub-step of the same run follows: its tile sort rides along (k_pair_all_sort)
      const bool presort_off = ctx->opt_no_presort != 0;  // (SDM_OPT_NO_PRESORT: A/B runs, tests)
      sort_ahead = split_one && !cell_path && !cfg->enable_breakup && !cfg->optimized_random &&
                   (n_sub + 1 < cfg->substeps || more_follow) && !ctx->graph_capture &&
                   A.rng_aff && !presort_off && A.rec != nullptr &&
                   (int64_t)grid_for(N, PCG_AFF_STRIDE) < PCG_AFF_TILES &&
                   sdm_shuffle_presort_ok(ctx, N, N);
      if (sort_ahead) {
        SortBuffers B;
        sdm_shuffle_sort_buffers(ctx, S.shuffle, N, &B);
        SortAhead X;
        memset(&X, 0, sizeof(X));
        X.events = B.events;
        X.toff = B.toff;
        X.jarr = B.jarr;
        X.loc = B.loc;
        X.n_bins = B.n_bins;
        X.n_tiles = B.n_tiles;
        X.p_length = st->cell_start + C;
        X.s_off = sdm_pcg_advance_host(rng_state, rng_inc, off);  // (off: the next draw's start)
        const dim3 grid((unsigned)(B.n_tiles + grid_for((N + 1) / 2, BIN_THREADS)));
        const dim3 big(BIN_THREADS);
#define PAIR_SORT(K) hipLaunchKernelGGL((k_pair_all_sort<K>), grid, big, B.lds_bytes, s, *cfg, A, X)
        switch (cfg->kernel) {
          case SDM_KERNEL_GOLOVIN: PAIR_SORT(SDM_KERNEL_GOLOVIN); break;
          case SDM_KERNEL_GEOMETRIC: PAIR_SORT(SDM_KERNEL_GEOMETRIC); break;
          case SDM_KERNEL_PARAMETERIZED: PAIR_SORT(SDM_KERNEL_PARAMETERIZED); break;
          case SDM_KERNEL_SIMPLE_GEOMETRIC: PAIR_SORT(SDM_KERNEL_SIMPLE_GEOMETRIC); break;
          case SDM_KERNEL_LINEAR: PAIR_SORT(SDM_KERNEL_LINEAR); break;
          default: PAIR_SORT(SDM_KERNEL_CONSTANT);
        }
#undef PAIR_SORT
        ctx->presorted.active = true;
        ctx->presorted.owner = st;
      } else {
        DISPATCH_PAIR(k_pair_all, dim3(grid_for((N + 1) / 2)));
      }
      LAUNCH_CHECK();
    } else {
      if (C > 1) {  // one cell: k_cells_adaptive does this part too
        PhaseScope ph(ctx, SDM_PHASE_CELLS_PRE);
        hipLaunchKernelGGL(k_cells_pre, dim3(grid_for(C)), blk, 0, s, *cfg, A);
        LAUNCH_CHECK();
      }
      if (!head_ok) {
        PhaseScope ph(ctx, SDM_PHASE_PAIR_PROB);
        DISPATCH_PAIR(k_pair_prob, dim3(grid_for((N + 1) / 2)));
        LAUNCH_CHECK();
      }
      // (one cell with at most 2048 partial minima: k_pair_update folds them itself)
      const bool fold = C == 1 && !sharded && !ctx->graph_capture && A.n_block_min <= 2048;
      A.fold_pre = fold ? (fill_pending ? 2 : 1) : 0;
      A.dt_left_pub = (const double *)(ctx->dscal + 6);
      if (!fold) {
        PhaseScope ph(ctx, SDM_PHASE_CELLS_ADAPTIVE);
        hipLaunchKernelGGL(k_cells_adaptive, dim3(grid_for(C)), blk, 0, s, *cfg, A,
                           C > 1 ? 0 : (fill_pending ? 2 : 1));
        LAUNCH_CHECK();
      }
      fill_pending = false;
      {
        PhaseScope ph(ctx, SDM_PHASE_PAIR_UPDATE);
        if (cfg->enable_breakup)
          hipLaunchKernelGGL(k_pair_update<true>, dim3(grid_for(P)), blk, 0, s, *cfg, A);
        else
          hipLaunchKernelGGL(k_pair_update<false>, dim3(grid_for(P)), blk, 0, s, *cfg, A);
        LAUNCH_CHECK();
      }
    }
    if (cfg->enable_breakup) {
      PhaseScope ph(ctx, SDM_PHASE_PAIR_UPDATE);
      const int64_t chunks = (A.list_cap + SDM_BLOCK - 1) / SDM_BLOCK;
      hipLaunchKernelGGL(k_resolve_dense, dim3((unsigned)(A.list_nl * chunks)), blk, 0, s, *cfg,
                         A);
      LAUNCH_CHECK();
      std::swap(A.list_count, A.list_count_next);  // it left the other set of counts cleared
    }
    if (sharded) {
      rc = shard_sync(cur);
      if (rc) return rc;
    }
    // (g) sanitize (particle_attributes.py:67-73), decided on the device by the healthy word
    {
      PhaseScope ph(ctx, SDM_PHASE_SANITIZE);
      // one adaptive cell: the kernel also ends the sub-step (collision.py:185-187: working
      // length = whole cell while dt_left > 0) and publishes the control block for the host
      CompactEpilogue epilogue = {nullptr, nullptr, nullptr, 0, nullptr, 0, 0};
      if (ctx->graph_capture) {
        epilogue.gwords = ctx->gwords;
        epilogue.advance = (uint64_t)(N + shift + P);
        epilogue.advance_b = cfg->enable_breakup ? (uint64_t)P : 0;
      }
      if (C == 1 && cfg->adaptive) {
        box_seq = ++ctx->poll_seq;
        epilogue.dt_left = st->dt_left;
        epilogue.slots = cfg->enable_breakup ? A.slots : nullptr;  // refused-breakup count
        epilogue.box = ctx->box_dev;
        epilogue.seq = box_seq;
        epilogue.dt_left_pub = (double *)(ctx->dscal + 6);
      }
      if (list_dead) {
        epilogue.dead_pos = A.dead_pos;
        epilogue.dead_count = A.dead_count;
        epilogue.dead_count_next = ctx->dead_ctr + ((ctx->dead_seq & 1) ^ 1) * 16;
        ctx->dead_seq += 1;
      }
      if (sort_ahead) {
        // the next build runs the compaction itself if a super-droplet died (k_bin_build2)
      } else {
        rc = sdm_compact_fused_async(ctx, S.compact, st->multiplicity, cur, N, N, st->ctl, S.cctl,
                                     C == 1 ? st->cell_start : nullptr, true, &epilogue);
        if (rc) return rc;
      }
      if (C > 1) sorted_host = -1;  // a compaction (decided on the device) un-sorts
    }
    ++n_sub;
    if (!cfg->adaptive && work_host >= 0) n_pairs += work_host / 2;
    if (cfg->adaptive && n_sub > max_substeps) {
      HIP_TRY(hipStreamSynchronize(s));
      return unbounded(n_sub, last_ctl);
    }
    if (cfg->adaptive) {
      // (h) collision.py:185-187 cut_working_length(adaptive_sdm_end(dt_left))
      n_pairs += work_host / 2;
      if (head_ok && more_follow) {  // ahead of the read-back (see launch_head)
        rc = launch_head();
        if (rc) return rc;
        head_done = true;
      }
      for (int attempt = 0; attempt < 2; ++attempt) {
        {
          PhaseScope ph(ctx, SDM_PHASE_ADAPTIVE_END);
          // the control block comes back through the polled box (publish_ctl), not a copy
          if (C > 1) {
            box_seq = ++ctx->poll_seq;
            rc = sdm_adaptive_end_async(ctx, st->dt_left, C, st->cell_start, S.end2, S.end2 + 1);
            if (rc) return rc;
            hipLaunchKernelGGL(k_set_work, one, one, 0, s, st->ctl, S.end2 + 1, ctx->box_dev,
                               box_seq);
            LAUNCH_CHECK();
          }
          rc = sdm_read_box(ctx, box_seq, last_ctl);
          if (rc) return rc;
        }
        have_ctl = true;
        work_host = last_ctl[CTL_WORK];
        if (C == 1 || last_ctl[CTL_SORTED] != 0) { sorted_host = C == 1 ? sorted_host : 1; break; }
        // a compaction happened in this sub-step: sort by cell first (particle_attributes.py
        // cell_start getter), then the end of the working range is taken from the new cell_start
        sorted_host = 0;
        hipLaunchKernelGGL(k_reset_work, one, one, 0, s, st->ctl);  // sanitize left work = valid
        LAUNCH_CHECK();
        rc = cond_sort(ctx, cfg, st, cur, S, &sorted_host);
        if (rc) return rc;
      }
    }
  }
  if (head_done) {  // the head launched last opens the next step: hand it over
    swaps -= 1;     // its buffer exchange counts there
    ctx->ahead.active = true;
    ctx->ahead.owner = st;
    ctx->ahead.off_before = head_off_before;
    ctx->ahead.off_b_before = head_off_b_before;
    ctx->ahead.s_rand = A.s_rand;
    ctx->ahead.s_rand_b = A.s_rand_b;
    ctx->ahead.rec = A.rec;
    ctx->ahead.rec_fmt = A.rec_fmt;
    ctx->ahead.ovf_head = A.ovf_head;
    ctx->ahead.ovf_next = A.ovf_next;
    ctx->ahead.cur = cur;
    ctx->ahead.alt = alt;
  }
  if (fill_pending) {  // no sub-step ran (nothing to work on): the fill still has to happen
    hipLaunchKernelGGL(k_fill_f64, dim3(grid_for(C)), blk, 0, s, st->dt_left, cfg->dt, C);
    LAUNCH_CHECK();
  }
  if (cfg->adaptive) {
    // collision.py:189-190 reset_working_length(); reset_cell_idx() (identity + sort)
    if (C == 1 && n_sub == 0) {  // (else the compaction's epilogue left work = valid)
      hipLaunchKernelGGL(k_reset_work, one, one, 0, s, st->ctl);
      LAUNCH_CHECK();
    }
    if (C > 1) {
      // by the cell_idx of the last counting sort.  (Local croupier only: under the global one
      // every sub-step sorts just its working range again, so after a cut cell_start describes
      // that range and not the whole state - the full counting sort it is, as in the reference)
      const bool grouped = sorted_host == 1 && cfg->croupier_local;
      hipLaunchKernelGGL(k_step_close, dim3(grid_for(C)), blk, 0, s, st->ctl, st->cell_idx, C,
                         S.gate_len, S.seg_size);
      LAUNCH_CHECK();
      if (grouped) {  // whole segments move (see k_reseg_sizes)
        PhaseScope ph(ctx, SDM_PHASE_SORT);
        hipLaunchKernelGGL(k_reseg_sizes, dim3(grid_for(C)), blk, 0, s, cur, st->cell_id,
                           st->cell_start, C, S.seg_size, S.seg_src);
        LAUNCH_CHECK();
        rc = sdm_cell_start_from_counts_async(ctx, S.seg_size, st->cell_start, C, S.gate_len + 1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_reseg_copy, dim3((unsigned)(C + 64)), blk, 0, s, alt, cur, S.seg_size,
                           S.seg_src, st->cell_start, C, N, st->ctl);
        LAUNCH_CHECK();
        { int64_t *t = cur; cur = alt; alt = t; }
        ++swaps;
        sorted_host = 1;
      } else {
        sorted_host = 0;
        rc = cond_sort(ctx, cfg, st, cur, S, &sorted_host, true);
        if (rc) return rc;
      }
    }
  }
  if (more_follow && cfg->adaptive && C > 1 && cfg->croupier_local && have_ctl &&
      sorted_host == 1 && max_cell >= 0) {
    ctx->carry.active = true;  // cells can only shrink within a call: max_cell stays a bound
    ctx->carry.owner = st;
    ctx->carry.valid = last_ctl[CTL_VALID];
    ctx->carry.max_cell = max_cell;
    ctx->carry.n_active_cells = n_active_cells;
  }
  if (A.slots && fold_counters) {
    hipLaunchKernelGGL(k_fold_counters, one, dim3(SDM_CNT_SLOTS), 0, s, A);
    LAUNCH_CHECK();
  }
  if (cfg->enable_breakup && more_follow) {  // (A.list_count: the set the last sub-step left clean)
    ctx->lists.active = true;
    ctx->lists.owner = st;
    ctx->lists.clean_set = A.list_count == S.list_count ? 0 : 1;
  }
  ctx->stats[SDM_STAT_SUBSTEPS] += n_sub;
  res->n_substeps = n_sub;
  res->idx_swapped = swaps & 1;
  res->rng_offset = off;
  res->rng_offset_breakup = off_b;
  res->valid_n_sd = -1;
  memset(res->ctl, 0, sizeof(res->ctl));
  if (read_back) {
    if (C == 1 && cfg->adaptive && have_ctl) {
      // the last sub-step's read-back already holds everything; only the working length was
      // reset since (k_reset_work: work = valid)
      memcpy(res->ctl, last_ctl, sizeof(last_ctl));
      res->ctl[CTL_WORK] = res->ctl[CTL_VALID];
    } else {
      HIP_TRY(hipMemcpyAsync(ctx->mailbox, st->ctl, sizeof(int64_t) * 8, hipMemcpyDeviceToHost,
                             s));
      HIP_TRY(hipStreamSynchronize(s));
      memcpy(res->ctl, ctx->mailbox, sizeof(res->ctl));
    }
    res->valid_n_sd = res->ctl[CTL_VALID];
    st->known_valid = res->valid_n_sd;
    if ((res->ctl[7] & SDM_CTL7_ERROR_MASK) == 2) (void)sdm_compact_rearm(ctx);  // the caller raises; the ctx stays usable
  } else {
    // one adaptive cell: the last sub-step's read-back told the valid length anyway - the next
    // step of the same run need not ask the device again
    st->known_valid = (C == 1 && cfg->adaptive && have_ctl) ? last_ctl[CTL_VALID] : -1;
  }
  // non-adaptive without read-back: the caller derives the pair count from its own length
  res->n_pairs = (cfg->adaptive || work_host >= 0) ? n_pairs : -1;
  st->rng_offset = off;
  st->rng_offset_breakup = off_b;
  return SDM_OK;
}

extern "C" int sdm_collision_step(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st,
                                  sdm_step_result *res, int flags) {
  ARG_TRY(ctx != nullptr);
  ctx->ahead.active = false;
  ctx->carry.active = false;
  ctx->lists.active = false;
  ctx->presorted.active = false;
  ctx->resort_backoff = 0;
  return collision_step(ctx, cfg, st, res, flags, true, false);
}

// n_steps consecutive time steps in one call (no host-side work between them): what
// `Particulator.run(n_steps)` amounts to when the collision dynamic is the only dynamic and nothing
// observes the intermediate states (PySDM/particulator.py:50-56).  state->idx / tmp_idx are
// exchanged in place whenever a step leaves the permutation in the other buffer; the result holds
// the totals, idx_swapped the parity over the whole run.
// ---- graph replay of a run (one cell, non-adaptive) ----------------------------------------------
// Such a time step is four or five launches with no host decision in between; at small n_sd the
// host's launch rate, not the GPU, sets the pace (2^16 super-droplets: 44 us per step for ~15 us of
// kernel time).  Two consecutive steps - after which the two permutation buffers are back in their
// roles - are captured once into a hipGraph and replayed; the only per-step inputs, the stream
// positions, live on the device (common.h: gwords).  The capture runs on a stream of the
// library's own (the caller's may be the legacy default stream, which cannot be captured),
// ordered against the caller's stream by events.
struct GraphKey {
  sdm_step_cfg cfg;
  const void *ptr[12];
  const void *arena;
  size_t arena_bytes;
};

static bool graph_eligible(sdm_ctx *ctx, const sdm_step_cfg *cfg, const sdm_step_state *st,
                           int64_t n_steps) {
  // Opt-in (SDM_GRAPH_REPLAY=1): measured on this runtime (ROCm 7.2, MI355X; profiles/README.md)
  // the replay is SLOWER than plain launches - 52.8 against 44.2 us per step at 2^16
  // super-droplets, 99 against 89 us at 2^20: a graph launch re-submits its kernel nodes one by
  // one with more overhead than hipLaunchKernel, and the step is bound by the dispatch latency
  // between dependent kernels (~11 us each at small sizes), not by the host.
  static const bool enabled = getenv("SDM_GRAPH_REPLAY") != nullptr;
  return enabled && cfg->n_cell == 1 && !cfg->adaptive && cfg->croupier_local &&
         !cfg->optimized_random && !ctx->timing && !st->cell_owned && n_steps >= 8 &&
         sdm_shuffle_can_split(cfg->n_sd, false);
}

static int graph_replay(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st,
                        int64_t n_double_steps) {
  GraphKey key;
  memset(&key, 0, sizeof(key));
  key.cfg = *cfg;
  const void *ptrs[12] = {st->idx, st->tmp_idx, st->multiplicity, st->attributes, st->cell_id,
                          st->cell_start, st->ctl, st->nm, st->collision_rate,
                          st->coalescence_rate, st->breakup_rate, st->gk_a};
  memcpy(key.ptr, ptrs, sizeof(ptrs));
  key.arena = ctx->arena;
  key.arena_bytes = ctx->arena_bytes;
  hipStream_t caller = ctx->stream;
  if (!ctx->own_stream) {
    HIP_TRY(hipStreamCreateWithFlags((hipStream_t *)&ctx->own_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags((hipEvent_t *)&ctx->own_event, hipEventDisableTiming));
    HIP_TRY(hipMalloc((void **)&ctx->gwords, sizeof(uint64_t) * 4));
  }
  hipStream_t own = (hipStream_t)ctx->own_stream;
  hipEvent_t ev = (hipEvent_t)ctx->own_event;
  const bool cached = ctx->graph_exec && ctx->graph_key_bytes == sizeof(key) &&
                      memcmp(ctx->graph_key, &key, sizeof(key)) == 0;
  if (!cached) {
    if (ctx->graph_exec) {
      (void)hipGraphExecDestroy((hipGraphExec_t)ctx->graph_exec);
      ctx->graph_exec = nullptr;
    }
    const uint64_t off = st->rng_offset, off_b = st->rng_offset_breakup;
    const int64_t known = st->known_valid;
    int64_t *const idx = st->idx, *const tmp = st->tmp_idx;
    hipGraph_t graph = nullptr;
    ctx->stream = own;
    ctx->graph_capture = true;
    int rc = SDM_OK;
    hipError_t e = hipStreamBeginCapture(own, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      for (int half = 0; half < 2 && rc == SDM_OK; ++half) {
        sdm_step_result one;
        rc = collision_step(ctx, cfg, st, &one, 0, false, true);
        if (rc == SDM_OK && one.idx_swapped) {
          int64_t *t = st->idx; st->idx = st->tmp_idx; st->tmp_idx = t;
        }
      }
      e = hipStreamEndCapture(own, &graph);
    }
    ctx->graph_capture = false;
    ctx->stream = caller;
    const bool roles_back = st->idx == idx && st->tmp_idx == tmp;
    st->idx = idx;
    st->tmp_idx = tmp;
    st->rng_offset = off;
    st->rng_offset_breakup = off_b;
    st->known_valid = known;
    if (rc != SDM_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    HIP_TRY(e);
    if (!roles_back) {  // (a step that does not exchange the buffers: not expected on this route)
      (void)hipGraphDestroy(graph);
      return 1;  // caller falls back to plain launches
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIP_TRY(e);
    ctx->graph_exec = exec;
    free(ctx->graph_key);
    ctx->graph_key = malloc(sizeof(key));
    if (!ctx->graph_key) return SDM_E_NOMEM;
    memcpy(ctx->graph_key, &key, sizeof(key));
    ctx->graph_key_bytes = sizeof(key);
  }
  // stream positions of the first replayed step, then the replays, between the caller's work
  uint64_t words[4] = {st->rng_offset, st->rng_offset_breakup, 0, 0};
  // (four rotating slots: an earlier call's asynchronous copy may not have run yet)
  int64_t *slot = ctx->mailbox + 48 + 4 * (ctx->graph_calls++ & 3);
  memcpy(slot, words, sizeof(words));
  HIP_TRY(hipEventRecord(ev, caller));
  HIP_TRY(hipStreamWaitEvent(own, ev, 0));
  HIP_TRY(hipMemcpyAsync(ctx->gwords, slot, sizeof(words), hipMemcpyHostToDevice, own));
  for (int64_t k = 0; k < n_double_steps; ++k)
    HIP_TRY(hipGraphLaunch((hipGraphExec_t)ctx->graph_exec, own));
  HIP_TRY(hipEventRecord(ev, own));
  HIP_TRY(hipStreamWaitEvent(caller, ev, 0));
  const int64_t N = cfg->n_sd, P = N / 2;
  const uint64_t sub = (uint64_t)(2 * n_double_steps) * (uint64_t)cfg->substeps;
  st->rng_offset += sub * (uint64_t)(N + P);
  if (cfg->enable_breakup) st->rng_offset_breakup += sub * (uint64_t)P;
  return SDM_OK;
}

// ---- cell-ordered working copy of a multi-step multi-cell run ------------------------------------
// The per-cell kernels gather one {multiplicity, mass, radius, velocity} record per super-droplet
// and sub-step BY ID - and ids are scattered over the whole population, so every gather is a
// 64-B sector miss somewhere in n_sd x 32 B (the gather phase was bound by the chip-wide miss rate:
// 32 % L2 hit rate, profiles/r02_pmc_kinematic2d_*).  Collisions never move a super-droplet to
// another cell.  So for the steps of a run that follow the first one (which leaves the state
// sorted by cell), the library works on a copy in which super-droplet i IS the one at position i
// of that sorted permutation: ids of a cell are consecutive, a workgroup's gathers and updates
// stay inside its cell's own window (4096 x 32 B = 128 KB: whole lines used, L2-resident), and the
// kernels are the same - they only see other pointers.  At the end of the run the copy is
// scattered back to the caller's columns and the permutation translated (idx = orig[idx']).
// `normalize`'s quirk - the factor of the cell of RAW super-droplet d for pair slot d - keeps
// reading the caller's cell_id (FusedArgs::cell_id_raw).
struct Relabel {
  bool on = false;
  int64_t n_entry = 0;  // live super-droplets at entry (labels [0, n_entry) are in use)
  int64_t *orig = nullptr, *perm_a = nullptr, *perm_b = nullptr, *multiplicity = nullptr,
          *cell_id = nullptr;
  double *attributes = nullptr;
  void *nm = nullptr;
  const char *arena = nullptr;  // where the copy was carved (the arena must not move meanwhile)
  sdm_step_state inner;
};

static size_t relabel_bytes(const sdm_step_cfg *cfg) {
  const size_t N = (size_t)cfg->n_sd;
  return carve_size(N * 8) * (5 + (size_t)cfg->n_attr) + carve_size(N * 32) + 4096;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_relabel_enter(const int64_t *__restrict__ idx, int64_t n_sd, int64_t n_attr, int64_t n_entry,
                const int64_t *__restrict__ multiplicity, const double *__restrict__ attributes,
                const int64_t *__restrict__ cell_id, const double *__restrict__ nm, int nm_wide,
                int64_t *__restrict__ orig, int64_t *__restrict__ perm_a,
                int64_t *__restrict__ perm_b, int64_t *__restrict__ mult_i,
                double *__restrict__ attr_i, int64_t *__restrict__ cid_i,
                double *__restrict__ nm_i) {
  const int64_t i = TID();
  if (i >= n_sd) return;
  const bool live = i < n_entry;
  const int64_t o = live ? idx[i] : 0;
  orig[i] = live ? o : n_sd;
  perm_a[i] = perm_b[i] = live ? i : n_sd;  // beyond the live length: the "removed" value
  mult_i[i] = live ? multiplicity[o] : 0;
  cid_i[i] = live ? cell_id[o] : 0;
  for (int64_t a = 0; a < n_attr; ++a) attr_i[a * n_sd + i] = live ? attributes[a * n_sd + o] : 0.0;
  if (nm_wide)
    ((double4 *)nm_i)[i] = live ? ((const double4 *)nm)[o] : make_double4(0, 0, 0, 0);
  else
    ((double2 *)nm_i)[i] = live ? ((const double2 *)nm)[o] : make_double2(0, 0);
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_relabel_exit(int64_t *__restrict__ idx, const int64_t *__restrict__ perm, int64_t n_sd,
               int64_t n_attr, int64_t n_entry, int64_t *__restrict__ multiplicity,
               double *__restrict__ attributes, double *__restrict__ nm, int nm_wide,
               const int64_t *__restrict__ orig, const int64_t *__restrict__ mult_i,
               const double *__restrict__ attr_i, const double *__restrict__ nm_i) {
  const int64_t i = TID();
  if (i >= n_sd) return;
  const int64_t v = perm[i];
  idx[i] = v < n_sd ? orig[v] : n_sd;
  if (i < n_entry) {
    const int64_t o = orig[i];
    multiplicity[o] = mult_i[i];
    for (int64_t a = 0; a < n_attr; ++a) attributes[a * n_sd + o] = attr_i[a * n_sd + i];
    if (nm_wide) ((double4 *)nm)[o] = ((const double4 *)nm_i)[i];
    else ((double2 *)nm)[o] = ((const double2 *)nm_i)[i];
  }
}

// after a step that left ctx->carry for `st` (sorted by cell, mirror current, per-cell route)
static int relabel_enter(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st, Relabel *R) {
  const int64_t N = cfg->n_sd;
  const size_t base = carve_size(layout(nullptr, cfg).total);
  Carver cv(ctx->arena + base);
  R->orig = cv.take<int64_t>((size_t)N);
  R->perm_a = cv.take<int64_t>((size_t)N);
  R->perm_b = cv.take<int64_t>((size_t)N);
  R->multiplicity = cv.take<int64_t>((size_t)N);
  R->cell_id = cv.take<int64_t>((size_t)N);
  R->attributes = cv.take<double>((size_t)(N * cfg->n_attr));
  R->nm = cv.take<double>((size_t)(4 * N));
  R->n_entry = ctx->carry.valid;
  R->arena = ctx->arena;
  hipLaunchKernelGGL(k_relabel_enter, dim3(grid_for(N)), dim3(SDM_BLOCK), 0, ctx->stream, st->idx,
                     N, cfg->n_attr, R->n_entry, st->multiplicity, st->attributes, st->cell_id,
                     (const double *)st->nm, mirror_is_wide(cfg) ? 1 : 0, R->orig, R->perm_a,
                     R->perm_b, R->multiplicity, R->attributes, R->cell_id, (double *)R->nm);
  LAUNCH_CHECK();
  R->inner = *st;
  R->inner.idx = R->perm_a;
  R->inner.tmp_idx = R->perm_b;
  R->inner.multiplicity = R->multiplicity;
  R->inner.attributes = R->attributes;
  R->inner.cell_id = R->cell_id;
  R->inner.nm = R->nm;
  ctx->cell_id_raw = st->cell_id_by_id ? st->cell_id_by_id : st->cell_id;
  // what the step just done left for the next one is as true of the copy
  ctx->carry.owner = &R->inner;
  if (ctx->lists.active) ctx->lists.owner = &R->inner;
  R->on = true;
  return SDM_OK;
}

static int relabel_exit(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st, Relabel *R) {
  if (!R->on) return SDM_OK;
  R->on = false;
  ctx->cell_id_raw = nullptr;
  const int64_t N = cfg->n_sd;
  hipLaunchKernelGGL(k_relabel_exit, dim3(grid_for(N)), dim3(SDM_BLOCK), 0, ctx->stream, st->idx,
                     R->inner.idx, N, cfg->n_attr, R->n_entry, st->multiplicity, st->attributes,
                     (double *)st->nm, mirror_is_wide(cfg) ? 1 : 0, R->orig, R->multiplicity,
                     R->attributes, (const double *)R->nm);
  LAUNCH_CHECK();
  st->rng_offset = R->inner.rng_offset;
  st->rng_offset_breakup = R->inner.rng_offset_breakup;
  st->known_valid = R->inner.known_valid;
  return SDM_OK;
}

extern "C" int sdm_collision_run(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *st,
                                 sdm_step_result *res, int flags, int64_t n_steps) {
  ARG_TRY(ctx && res && st && n_steps >= 0);
  ctx->ahead.active = false;
  ctx->carry.active = false;
  ctx->lists.active = false;
  ctx->presorted.active = false;
  ctx->resort_backoff = 0;
  sdm_step_result total;
  memset(&total, 0, sizeof(total));
  total.valid_n_sd = -1;
  total.rng_offset = st->rng_offset;
  total.rng_offset_breakup = st->rng_offset_breakup;
  bool pairs_known = true;
  const bool replay = cfg && graph_eligible(ctx, cfg, st, n_steps);
  // cell-ordered working copy (above): multi-cell adaptive runs of three steps or more on the
  // per-cell route (SDM_OPT_NO_CELL_COPY switches it off: measurements, tests).  Sharded runs too: labels
  // are a process's own business - only positions and per-cell numbers cross processes
  const bool copy_enabled = !ctx->opt_no_cell_copy;
  Relabel relabel;
  const bool copy_wanted = copy_enabled && cfg && cfg->n_cell > 1 && cfg->adaptive &&
                           cfg->croupier_local && st->nm && n_steps >= 3 && !ctx->graph_capture;
  if (copy_wanted) {  // all scratch up front: the arena must not move once the copy lives in it
    const int rc = sdm_reserve(ctx, carve_size(layout(nullptr, cfg).total) + relabel_bytes(cfg));
    if (rc) return rc;
  }
  sdm_step_state *use = st;
  for (int64_t step = 0; step < n_steps; ++step) {
    if (replay && step == 1) {
      // steps 1 .. 2k replayed two at a time; step 0 (allocations, a fresh control block) and
      // the tail (read-back, counters folded) launch as usual
      const int64_t doubles = (n_steps - 2) / 2;
      const int rc = graph_replay(ctx, cfg, st, doubles);
      if (rc < 0) return rc;
      if (rc == 0) {
        step += 2 * doubles - 1;
        total.n_substeps += 2 * doubles * cfg->substeps;
        pairs_known = false;  // (counted on the device: control word 5)
        total.rng_offset = st->rng_offset;
        total.rng_offset_breakup = st->rng_offset_breakup;
        continue;
      }
    }
    sdm_step_result one;
    const bool last = step == n_steps - 1;
    // read the control block back only after the last step
    int rc = collision_step(ctx, cfg, use, &one, (last ? (flags & 1) : 0) |
                                                 (step == 0 ? (flags & 2) : 0), last, !last);
    if (rc) {
      ctx->ahead.active = false;
      ctx->carry.active = false;
      (void)relabel_exit(ctx, cfg, st, &relabel);
      return rc;
    }
    if (one.idx_swapped) {
      int64_t *t = use->idx;
      use->idx = use->tmp_idx;
      use->tmp_idx = t;
      if (use == st) total.idx_swapped ^= 1;  // (the copy's buffers are the library's own)
    }
    if (copy_wanted && !relabel.on && use == st && !last && ctx->carry.active &&
        ctx->carry.owner == (const void *)st && ctx->carry.max_cell >= 0 &&
        ctx->carry.max_cell <= CELL_CAP) {
      rc = relabel_enter(ctx, cfg, st, &relabel);
      if (rc) return rc;
      use = &relabel.inner;
    }
    if (relabel.on && relabel.arena != ctx->arena) {  // (cannot happen: reserved up front)
      sdm_set_error("the scratch arena moved under the cell-ordered working copy");
      ctx->cell_id_raw = nullptr;
      return SDM_E_HIP;
    }
    if (last && relabel.on) {
      // (the read-back of the last step is through; the caller's columns are made current again)
      rc = relabel_exit(ctx, cfg, st, &relabel);
      if (rc) return rc;
      use = st;
    }
    total.n_substeps += one.n_substeps;
    if (one.n_pairs < 0) pairs_known = false; else total.n_pairs += one.n_pairs;
    total.valid_n_sd = one.valid_n_sd;
    memcpy(total.ctl, one.ctl, sizeof(total.ctl));
    total.rng_offset = one.rng_offset;
    total.rng_offset_breakup = one.rng_offset_breakup;
  }
  if (!pairs_known) total.n_pairs = -1;
  *res = total;
  return SDM_OK;
}
