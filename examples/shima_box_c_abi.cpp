// The Shima-2009 coalescence box driven through the C ABI alone (include/sdm_hip.h): no Python,
// no torch -- what a binding in another host language (cgo, JNI, ...) would do.  Device memory
// comes from the HIP runtime; the library owns nothing but its opaque context.
//
//   shima_box_c_abi <input> <output>
//   input  (binary): int64 n_sd, n_steps; double dt, dv, b; uint64 rng[4] (state hi/lo, inc hi/lo
//                    of numpy.random.PCG64(seed)); int64 multiplicity[n_sd]; double mass[n_sd]
//   output (binary): int64 n_live; int64 idx[n_sd]; int64 multiplicity[n_sd]; double mass[n_sd]
//
// tests/test_hip_parity.py runs it next to the Python route on the same input and compares the
// results bit for bit.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/sdm_hip.h"

#define HIP_OK(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));             \
      return 2;                                                                   \
    }                                                                             \
  } while (0)
#define SDM_OK_(expr)                                                             \
  do {                                                                            \
    if ((expr) != SDM_OK) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #expr, sdm_last_error());                  \
      return 3;                                                                   \
    }                                                                             \
  } while (0)

template <typename T>
static T *device_array(size_t n) {
  T *p = nullptr;
  if (hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)) != hipSuccess) return nullptr;
  (void)hipMemset(p, 0, sizeof(T) * (n ? n : 1));
  return p;
}

int main(int argc, char **argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s <input> <output>\n", argv[0]);
    return 1;
  }
  FILE *in = std::fopen(argv[1], "rb");
  if (!in) return 1;
  int64_t n_sd = 0, n_steps = 0;
  double dt = 0, dv = 0, b = 0;
  uint64_t rng[4];
  if (std::fread(&n_sd, 8, 1, in) != 1 || std::fread(&n_steps, 8, 1, in) != 1 ||
      std::fread(&dt, 8, 1, in) != 1 || std::fread(&dv, 8, 1, in) != 1 ||
      std::fread(&b, 8, 1, in) != 1 || std::fread(rng, 8, 4, in) != 4)
    return 1;
  std::vector<int64_t> multiplicity(n_sd), idx(n_sd);
  std::vector<double> mass(n_sd);
  if (std::fread(multiplicity.data(), 8, n_sd, in) != (size_t)n_sd ||
      std::fread(mass.data(), 8, n_sd, in) != (size_t)n_sd)
    return 1;
  std::fclose(in);
  for (int64_t i = 0; i < n_sd; ++i) idx[i] = i;

  sdm_ctx *ctx = nullptr;
  SDM_OK_(sdm_ctx_create(&ctx, 0));
  if (sdm_abi_version() != 1) return 3;

  // caller-owned state, exactly the reference's Storages (SoA, int64 / float64)
  sdm_step_state st;
  std::memset(&st, 0, sizeof(st));
  st.idx = device_array<int64_t>(n_sd);
  st.tmp_idx = device_array<int64_t>(n_sd);
  st.multiplicity = device_array<int64_t>(n_sd);
  st.attributes = device_array<double>(n_sd);  // one extensive attribute: signed water mass
  st.cell_id = device_array<int64_t>(n_sd);    // all zero: one cell
  st.cell_idx = device_array<int64_t>(1);
  st.cell_start = device_array<int64_t>(2);
  st.dt_left = device_array<double>(1);
  st.stats_dt_min = device_array<double>(1);
  st.stats_n_substep = device_array<int64_t>(1);
  st.collision_rate = device_array<int64_t>(1);
  st.collision_rate_deficit = device_array<int64_t>(1);
  st.coalescence_rate = device_array<int64_t>(1);
  st.ctl = device_array<int64_t>(8);
  st.nm = device_array<int64_t>(4 * n_sd);     // the library's {multiplicity, mass} mirror
  st.known_valid = -1;
  HIP_OK(hipMemcpy(st.idx, idx.data(), 8 * n_sd, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(st.multiplicity, multiplicity.data(), 8 * n_sd, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(st.attributes, mass.data(), 8 * n_sd, hipMemcpyHostToDevice));
  const int64_t ctl[8] = {n_sd, n_sd, 0, 1, 0, 0, 0, 0};  // valid, working, sorted, healthy, ..
  HIP_OK(hipMemcpy(st.ctl, ctl, sizeof(ctl), hipMemcpyHostToDevice));

  // Coalescence(collision_kernel=Golovin(b), adaptive=False) in Box(dt, dv)
  sdm_step_cfg cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.n_sd = n_sd;
  cfg.n_cell = 1;
  cfg.n_attr = 1;
  cfg.dt = dt;
  cfg.dv = dv;
  cfg.dt_min = 0.1;
  cfg.dt_max = dt < 100.0 ? dt : 100.0;
  cfg.adaptive = 0;
  cfg.substeps = 1;
  cfg.croupier_local = 1;
  cfg.kernel = SDM_KERNEL_GOLOVIN;
  cfg.kernel_param[0] = b;
  cfg.ec = SDM_EC_CONST;
  cfg.ec_param[0] = 1.0;
  cfg.frag_nfmax = -1.0;
  cfg.rho_w = 1000.0;
  cfg.sgm_w = 0.072;
  cfg.max_multiplicity = INT64_MAX / 200000;
  std::memcpy(cfg.rng_state_inc, rng, sizeof(rng));

  sdm_step_result res;
  SDM_OK_(sdm_collision_run(ctx, &cfg, &st, &res, SDM_STEP_READ_BACK | SDM_STEP_FRESH_CTL,
                            n_steps));
  SDM_OK_(sdm_ctx_synchronize(ctx));

  HIP_OK(hipMemcpy(idx.data(), st.idx, 8 * n_sd, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(multiplicity.data(), st.multiplicity, 8 * n_sd, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(mass.data(), st.attributes, 8 * n_sd, hipMemcpyDeviceToHost));
  FILE *out = std::fopen(argv[2], "wb");
  if (!out) return 1;
  std::fwrite(&res.valid_n_sd, 8, 1, out);
  std::fwrite(idx.data(), 8, n_sd, out);
  std::fwrite(multiplicity.data(), 8, n_sd, out);
  std::fwrite(mass.data(), 8, n_sd, out);
  std::fclose(out);
  std::printf("%lld steps, %lld of %lld super-droplets live, %lld candidate pairs\n",
              (long long)res.n_substeps, (long long)res.valid_n_sd, (long long)n_sd,
              (long long)res.n_pairs);
  SDM_OK_(sdm_ctx_destroy(ctx));
  return 0;
}
