// ctx.hip -- context, error reporting, PCG64 stream fill, Storage element-wise ops, reductions
#include <stdarg.h>
#include <stdlib.h>
#include <unistd.h>

#include <cstring>
#include "common.h"

static thread_local char g_err[512] = "";

void sdm_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *sdm_last_error(void) { return g_err; }
extern "C" int sdm_abi_version(void) { return 1; }

static int ctx_allocate(sdm_ctx *ctx) {
  HIP_TRY(hipMalloc((void **)&ctx->pcg_tab, sizeof(u128) * 128));
  HIP_TRY(hipMalloc((void **)&ctx->pcg_aff, sizeof(u128) * 2 * (PCG_AFF_SMALL + PCG_AFF_TILES)));
  HIP_TRY(hipHostMalloc((void **)&ctx->mailbox, sizeof(int64_t) * SDM_MAILBOX_WORDS,
                        hipHostMallocMapped | hipHostMallocCoherent));
  memset(ctx->mailbox, 0, sizeof(int64_t) * SDM_MAILBOX_WORDS);
  if (const char *delay = getenv("SDM_DEBUG_BOX_DELAY_US")) ctx->debug_box_delay_us = atoi(delay);
  // process-wide defaults of three A/B options (sdm_hip.h)
  if (const char *fmt = getenv("SDM_REC_FORMAT")) ctx->opt_records = !strcmp(fmt, "records");
  ctx->opt_no_presort = getenv("SDM_NO_PRESORT") != nullptr;
  if (const char *copy = getenv("SDM_CELL_COPY")) ctx->opt_no_cell_copy = copy[0] == '0';
  if (const char *shape = getenv("SDM_CELL_SHAPE")) {
    const int v = atoi(shape);
    if (v >= SDM_CELL_SHAPE_AUTO && v <= SDM_CELL_SHAPE_256) ctx->opt_cell_shape = v;
  }
  HIP_TRY(hipHostGetDevicePointer((void **)&ctx->box_dev, ctx->mailbox + SDM_BOX, 0));
  HIP_TRY(hipMalloc((void **)&ctx->dscal, sizeof(int64_t) * 16));
  HIP_TRY(hipMemset(ctx->dscal, 0, sizeof(int64_t) * 16));
  HIP_TRY(hipMalloc((void **)&ctx->cnt_slots, sizeof(int64_t) * SDM_CNT_SLOTS * SDM_CNT_STRIDE));
  HIP_TRY(hipMemset(ctx->cnt_slots, 0, sizeof(int64_t) * SDM_CNT_SLOTS * SDM_CNT_STRIDE));
  HIP_TRY(hipMalloc((void **)&ctx->dead_pos, sizeof(int64_t) * SDM_DEAD_LIST_CAP));
  HIP_TRY(hipMalloc((void **)&ctx->dead_ctr, sizeof(unsigned long long) * 32));
  HIP_TRY(hipMemset(ctx->dead_ctr, 0, sizeof(unsigned long long) * 32));
  return SDM_OK;
}

extern "C" int sdm_ctx_create(sdm_ctx **out, int device) {
  ARG_TRY(out != nullptr);
  HIP_TRY(hipSetDevice(device));
  sdm_ctx *ctx = new sdm_ctx();
  memset(ctx, 0, sizeof(*ctx));
  ctx->device = device;
  ctx->stream = nullptr;
  const int rc = ctx_allocate(ctx);
  if (rc != SDM_OK) {  // release whatever was allocated before the failing call
    (void)sdm_ctx_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return SDM_OK;
}

// waits until a kernel published sequence number `seq` (publish_ctl); the stream is asked now and
// then, so that a failed launch or a fault ends the wait with an error instead of a hang
int sdm_wait_box(sdm_ctx *ctx, int64_t seq) {
  const int64_t *flag = ctx->mailbox + SDM_BOX + (seq & 1) * SDM_BOX_STRIDE + 8;
  // test knob: a host that is slower than the device by about a sub-step (the situation in which
  // a single-slot box would be overwritten by the sub-step launched ahead)
  if (ctx->debug_box_delay_us > 0) usleep((useconds_t)ctx->debug_box_delay_us);
  for (uint64_t spins = 1;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SDM_OK;
    if ((spins & 0x3fff) == 0) {
      const hipError_t e = hipStreamQuery(ctx->stream);
      if (e == hipSuccess) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SDM_OK;
        sdm_set_error("control block was not published although the stream is idle");
        return SDM_E_HIP;
      }
      if (e != hipErrorNotReady) HIP_TRY(e);
    }
    __builtin_ia32_pause();
  }
}

int sdm_read_box(sdm_ctx *ctx, int64_t seq, int64_t out[8]) {
  const int rc = sdm_wait_box(ctx, seq);
  if (rc) return rc;
  const int64_t *slot = ctx->mailbox + SDM_BOX + (seq & 1) * SDM_BOX_STRIDE;
  // the data words may arrive after the sequence word (common.h: publish_ctl): every word must
  // carry this publication's tag
  const uint64_t tag = (uint64_t)seq & SDM_BOX_TAG_MASK;
  for (uint64_t spins = 1;; ++spins) {
    bool whole = true;
    for (int w = 0; w < 8; ++w) {
      const uint64_t x = (uint64_t)__atomic_load_n(slot + w, __ATOMIC_ACQUIRE);
      whole = whole && (x >> SDM_BOX_TAG_SHIFT) == tag;
      out[w] = (int64_t)(x & SDM_BOX_VALUE_MASK);
    }
    if (whole) break;
    if (spins > (1ull << 28)) {  // (seconds: the words of a publication are microseconds apart)
      sdm_set_error("control block %lld: its words never became whole", (long long)seq);
      return SDM_E_HIP;
    }
    __builtin_ia32_pause();
  }
  if (__atomic_load_n(slot + 8, __ATOMIC_ACQUIRE) != seq) {
    sdm_set_error("control block %lld was overwritten while it was read", (long long)seq);
    return SDM_E_HIP;
  }
  return SDM_OK;
}

extern "C" int sdm_ctx_destroy(sdm_ctx *ctx) {
  if (!ctx) return SDM_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)sdm_comm_destroy(ctx);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->pcg_tab) (void)hipFree(ctx->pcg_tab);
  if (ctx->pcg_aff) (void)hipFree(ctx->pcg_aff);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  if (ctx->dscal) (void)hipFree(ctx->dscal);
  if (ctx->cnt_slots) (void)hipFree(ctx->cnt_slots);
  if (ctx->dead_pos) (void)hipFree(ctx->dead_pos);
  if (ctx->dead_ctr) (void)hipFree(ctx->dead_ctr);
  if (ctx->rain_carry) (void)hipFree(ctx->rain_carry);
  if (ctx->graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)ctx->graph_exec);
  free(ctx->graph_key);
  if (ctx->gwords) (void)hipFree(ctx->gwords);
  if (ctx->own_event) (void)hipEventDestroy((hipEvent_t)ctx->own_event);
  if (ctx->own_stream) (void)hipStreamDestroy((hipStream_t)ctx->own_stream);
  if (ctx->ev) {
    for (int i = 0; i < SDM_MAX_EVENTS; ++i) (void)hipEventDestroy(ctx->ev[i]);
    delete[] ctx->ev;
    delete[] ctx->ev_phase;
  }
  delete ctx;
  return SDM_OK;
}

// elapsed times of the recorded (begin, end) pairs into phase_ms / phase_count; empties the pool
static int resolve_events(sdm_ctx *ctx) {
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i + 1 < ctx->n_ev; i += 2) {
    const int phase = ctx->ev_phase[i / 2];
    if (phase < 0 || phase >= SDM_N_PHASES) continue;
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, ctx->ev[i], ctx->ev[i + 1]));
    ctx->phase_ms[phase] += t;
    ctx->phase_count[phase] += 1;
  }
  ctx->n_ev = 0;
  return SDM_OK;
}

void sdm_phase_begin(sdm_ctx *ctx, int phase) {
  // pool full (a long run in one call): resolve what is there - timing mode is a diagnostic pass
  if (ctx->n_ev + 2 > SDM_MAX_EVENTS && resolve_events(ctx) != SDM_OK) return;
  ctx->ev_phase[ctx->n_ev / 2] = phase;
  (void)hipEventRecord(ctx->ev[ctx->n_ev], ctx->stream);
  ctx->n_ev += 1;
}

void sdm_phase_end(sdm_ctx *ctx) {
  if (!(ctx->n_ev & 1)) return;  // begin was dropped (pool full)
  (void)hipEventRecord(ctx->ev[ctx->n_ev], ctx->stream);
  ctx->n_ev += 1;
}

extern "C" int sdm_ctx_set_option(sdm_ctx *ctx, int option, int64_t value) {
  ARG_TRY(ctx != nullptr);
  if (option == SDM_OPT_RESORT) {
    ARG_TRY(value == SDM_RESORT_AUTO || value == SDM_RESORT_COUNTING_SORT ||
            value == SDM_RESORT_ALWAYS_ASK);
    ctx->opt_resort = (int)value;
    return SDM_OK;
  }
  if (option == SDM_OPT_MAX_SUBSTEPS) {
    ARG_TRY(value >= 0);
    ctx->opt_max_substeps = value;
    return SDM_OK;
  }
  if (option == SDM_OPT_REC_FORMAT || option == SDM_OPT_NO_PRESORT ||
      option == SDM_OPT_NO_CELL_COPY) {
    ARG_TRY(value == 0 || value == 1);
    // (a tile sort that rode in the last pair kernel, a sub-step launched ahead: both belong to
    // the format they were made for - none is pending between calls of a caller that changes it,
    // and the entry points drop what they cannot use)
    if (option == SDM_OPT_REC_FORMAT) ctx->opt_records = (int)value;
    else if (option == SDM_OPT_NO_PRESORT) ctx->opt_no_presort = (int)value;
    else ctx->opt_no_cell_copy = (int)value;
    ctx->presorted.active = false;
    ctx->build_resident = 0;  // (the occupancy of the build kernel depends on the format)
    return SDM_OK;
  }
  if (option == SDM_OPT_CELL_SHAPE) {
    ARG_TRY(value >= SDM_CELL_SHAPE_AUTO && value <= SDM_CELL_SHAPE_256);
    ctx->opt_cell_shape = (int)value;
    return SDM_OK;
  }
  sdm_set_error("sdm_ctx_set_option: unknown option %d", option);
  return SDM_E_ARG;
}

extern "C" int sdm_ctx_read_stats(sdm_ctx *ctx, int64_t *stats, int clear) {
  ARG_TRY(ctx && stats);
  for (int k = 0; k < SDM_N_STATS; ++k) {
    stats[k] = ctx->stats[k];
    if (clear) ctx->stats[k] = 0;
  }
  return SDM_OK;
}

extern "C" int sdm_ctx_set_timing(sdm_ctx *ctx, int enable) {
  ARG_TRY(ctx != nullptr);
  if (enable && !ctx->ev) {
    ctx->ev = new hipEvent_t[SDM_MAX_EVENTS];
    ctx->ev_phase = new int[SDM_MAX_EVENTS / 2];
    for (int i = 0; i < SDM_MAX_EVENTS; ++i) HIP_TRY(hipEventCreate(&ctx->ev[i]));
  }
  ctx->timing = enable != 0;
  return SDM_OK;
}

extern "C" int sdm_ctx_read_timing(sdm_ctx *ctx, double *ms, int64_t *count) {
  ARG_TRY(ctx && ms && count);
  const int rc = resolve_events(ctx);
  if (rc) return rc;
  for (int p = 0; p < SDM_N_PHASES; ++p) {
    ms[p] = ctx->phase_ms[p];
    count[p] = ctx->phase_count[p];
    ctx->phase_ms[p] = 0;
    ctx->phase_count[p] = 0;
  }
  return SDM_OK;
}

extern "C" const char *sdm_phase_name(int phase) {
  static const char *names[SDM_N_PHASES] = {
      "counting_sort", "pcg64_fill", "shuffle_clear", "shuffle_build", "shuffle_trace",
      "tail_copy", "cells_pre", "pair_prob", "cells_adaptive", "pair_update", "sanitize",
      "adaptive_end"};
  return (phase >= 0 && phase < SDM_N_PHASES) ? names[phase] : "?";
}

extern "C" int sdm_ctx_set_stream(sdm_ctx *ctx, void *hip_stream) {
  ARG_TRY(ctx != nullptr);
  ctx->stream = (hipStream_t)hip_stream;
  return SDM_OK;
}

extern "C" int sdm_ctx_synchronize(sdm_ctx *ctx) {
  ARG_TRY(ctx != nullptr);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SDM_OK;
}

int sdm_reserve(sdm_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->arena_bytes) return SDM_OK;
  // growing the arena must not race with work still using the old one
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->arena) HIP_TRY(hipFree(ctx->arena));
  ctx->arena = nullptr;
  ctx->arena_bytes = 0;
  size_t want = bytes + bytes / 4 + (1 << 20);
  hipError_t e = hipMalloc((void **)&ctx->arena, want);
  if (e != hipSuccess) {
    sdm_set_error("scratch arena: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    return SDM_E_NOMEM;
  }
  ctx->arena_bytes = want;
  return SDM_OK;
}

// ---- PCG64 -----------------------------------------------------------------------------
__global__ void k_pcg_table(u128 *tab, u128 inc) {
  u128 cur_mult = pcg_mult(), cur_plus = inc;
  for (int b = 0; b < 64; ++b) {
    tab[2 * b] = cur_mult;
    tab[2 * b + 1] = cur_plus;
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
  }
}

// the affine maps of ctx->pcg_aff: the per-bit maps of `tab` commute, so their composition over
// the set bits of a distance is the jump by that distance
__global__ void __launch_bounds__(SDM_BLOCK) k_pcg_affine(u128 *aff, const u128 *tab) {
  const int64_t k = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (k >= PCG_AFF_SMALL + PCG_AFF_TILES) return;
  uint64_t delta = k < PCG_AFF_SMALL ? (uint64_t)k
                                     : (uint64_t)(k - PCG_AFF_SMALL) * PCG_AFF_STRIDE;
  u128 mult = 1, plus = 0;
  for (int b = 0; delta; ++b, delta >>= 1)
    if (delta & 1) {
      mult *= tab[2 * b];
      plus = plus * tab[2 * b] + tab[2 * b + 1];
    }
  aff[2 * k] = mult;
  aff[2 * k + 1] = plus;
}

int sdm_pcg_prepare(sdm_ctx *ctx, const uint64_t state_inc[4]) {
  const u128 inc = (((u128)state_inc[2]) << 64) | state_inc[3];
  if (!ctx->tab_valid || ctx->tab_inc != inc) {
    hipLaunchKernelGGL(k_pcg_table, dim3(1), dim3(1), 0, ctx->stream, ctx->pcg_tab, inc);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pcg_affine, dim3(grid_for(PCG_AFF_SMALL + PCG_AFF_TILES)),
                       dim3(SDM_BLOCK), 0, ctx->stream, ctx->pcg_aff, ctx->pcg_tab);
    LAUNCH_CHECK();
    ctx->tab_inc = inc;
    ctx->tab_valid = true;
  }
  return SDM_OK;
}

// host-side jump-ahead (pcg_advance_lcg_128)
u128 sdm_pcg_advance_host(u128 state, u128 inc, uint64_t delta) {
  u128 acc_mult = 1, acc_plus = 0, cur_mult = pcg_mult(), cur_plus = inc;
  while (delta > 0) {
    if (delta & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta >>= 1;
  }
  return acc_mult * state + acc_plus;
}

#define PCG_ELEMS 4
// out[i] = draw number (offset + i); `s_off` = generator state after `offset` draws
__global__ void __launch_bounds__(SDM_BLOCK)
k_pcg_fill(double *__restrict__ out, int64_t n, u128 s_off, u128 inc,
           const u128 *__restrict__ tab) {
  __shared__ u128 s_blk;
  const int64_t blk_first = (int64_t)blockIdx.x * (SDM_BLOCK * PCG_ELEMS);
  if (threadIdx.x == 0) s_blk = pcg_jump(s_off, tab, (uint64_t)blk_first);
  __syncthreads();
  u128 state = pcg_jump(s_blk, tab, (uint64_t)threadIdx.x * PCG_ELEMS);
  const int64_t first = blk_first + (int64_t)threadIdx.x * PCG_ELEMS;
  const u128 mult = pcg_mult();
  double v[PCG_ELEMS];
#pragma unroll
  for (int e = 0; e < PCG_ELEMS; ++e) {
    state = state * mult + inc;
    v[e] = pcg_output(state);
  }
  if (first + PCG_ELEMS <= n && (((uintptr_t)(out + first)) & 15) == 0) {
    double2 *o = (double2 *)(out + first);
    o[0] = make_double2(v[0], v[1]);
    o[1] = make_double2(v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < PCG_ELEMS; ++e)
      if (first + e < n) out[first + e] = v[e];
  }
}

int sdm_pcg_fill_async(sdm_ctx *ctx, double *out, int64_t n, const uint64_t state_inc[4],
                       uint64_t offset) {
  if (n <= 0) return SDM_OK;
  int rc = sdm_pcg_prepare(ctx, state_inc);
  if (rc) return rc;
  const u128 st = (((u128)state_inc[0]) << 64) | state_inc[1];
  const u128 inc = (((u128)state_inc[2]) << 64) | state_inc[3];
  const u128 s_off = sdm_pcg_advance_host(st, inc, offset);
  hipLaunchKernelGGL(k_pcg_fill, dim3(grid_for(n, SDM_BLOCK * PCG_ELEMS)), dim3(SDM_BLOCK), 0,
                     ctx->stream, out, n, s_off, inc, ctx->pcg_tab);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_pcg64_uniform(sdm_ctx *ctx, double *out, int64_t n,
                                 const uint64_t state_inc[4], uint64_t offset) {
  ARG_TRY(ctx && state_inc && n >= 0 && (out || n == 0));
  return sdm_pcg_fill_async(ctx, out, n, state_inc, offset);
}

// ---- element-wise ----------------------------------------------------------------------
__device__ __forceinline__ double py_mod_f64(double a, double b) {
  double r = fmod(a, b);
  if (r != 0 && ((r < 0) != (b < 0))) r += b;
  return r;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_ew_f64(int op, double *out, const double *a, const double *b, double s, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double x = a ? a[i] : 0.0;
  const double y = b ? b[i] : s;
  double r;
  switch (op) {
    case SDM_EW_ADD: r = x + y; break;
    case SDM_EW_SUB: r = x - y; break;
    case SDM_EW_MUL: r = x * y; break;
    case SDM_EW_DIV: r = x / y; break;
    case SDM_EW_POW: {
      const double sg = (x > 0) - (x < 0);
      // exponent 2 (Geometric kernel, Berry Ec): the exact square, as numpy / libm give it
      r = (x != x) ? x : sg * (s == 2.0 ? x * x : sdm_pow(fabs(x), s));
      break;
    }
    case SDM_EW_DIV_IF_NOT_ZERO: r = (y != 0.0) ? x / y : x; break;
    case SDM_EW_FLOOR: r = floor(x); break;
    case SDM_EW_EXP: r = sdm_exp(x); break;
    case SDM_EW_ABS: r = fabs(x); break;
    case SDM_EW_FILL: r = y; break;
    case SDM_EW_ADD_MUL: r = x + s * b[i]; break;
    case SDM_EW_MOD: r = py_mod_f64(x, y); break;
    default: r = x;
  }
  out[i] = r;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_math_eval(int fn, double *out, const double *a, const double *b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double x = a[i];
  double r;
  switch (fn) {
    case SDM_MATH_EXP: r = sdm_exp(x); break;
    case SDM_MATH_LOG: r = sdm_log(x); break;
    case SDM_MATH_POW: r = sdm_pow(x, b[i]); break;
    case SDM_MATH_SINH: r = sdm_sinh(x); break;
    case SDM_MATH_ASINH: r = sdm_asinh(x); break;
    case SDM_MATH_ATANH: r = sdm_atanh(x); break;
    case SDM_MATH_ERF: r = sdm_erf(x); break;
    default: r = sdm_log1p(x);
  }
  out[i] = r;
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_ew_i64(int op, int64_t *out, const int64_t *a, const int64_t *b, int64_t s, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int64_t x = a ? a[i] : 0;
  const int64_t y = b ? b[i] : s;
  int64_t r;
  switch (op) {
    case SDM_EW_ADD: r = x + y; break;
    case SDM_EW_SUB: r = x - y; break;
    case SDM_EW_MUL: r = x * y; break;
    case SDM_EW_ABS: r = x < 0 ? -x : x; break;
    case SDM_EW_FILL: r = y; break;
    case SDM_EW_MOD: {
      r = x % y;
      if (r != 0 && ((r < 0) != (y < 0))) r += y;
      break;
    }
    default: r = x;
  }
  out[i] = r;
}

extern "C" int sdm_elementwise_f64(sdm_ctx *ctx, int op, double *out, const double *a,
                                   const double *b, double scalar, int64_t n) {
  ARG_TRY(ctx && n >= 0 && (out || n == 0));
  ARG_TRY(op >= 0 && op <= SDM_EW_MOD);
  ARG_TRY(op != SDM_EW_ADD_MUL || b != nullptr);
  if (n == 0) return SDM_OK;
  hipLaunchKernelGGL(k_ew_f64, dim3(grid_for(n)), dim3(SDM_BLOCK), 0, ctx->stream, op, out, a,
                     b, scalar, n);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_elementwise_i64(sdm_ctx *ctx, int op, int64_t *out, const int64_t *a,
                                   const int64_t *b, int64_t scalar, int64_t n) {
  ARG_TRY(ctx && n >= 0 && (out || n == 0));
  ARG_TRY(op == SDM_EW_ADD || op == SDM_EW_SUB || op == SDM_EW_MUL || op == SDM_EW_ABS ||
          op == SDM_EW_FILL || op == SDM_EW_MOD);
  if (n == 0) return SDM_OK;
  hipLaunchKernelGGL(k_ew_i64, dim3(grid_for(n)), dim3(SDM_BLOCK), 0, ctx->stream, op, out, a,
                     b, scalar, n);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---- min / max reduction (np.amin / np.amax: NaN propagates) ------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_reduce_f64(int kind, const double *__restrict__ a, int64_t n, double *__restrict__ partial,
             int *__restrict__ has_nan) {
  __shared__ double sm[SDM_BLOCK / SDM_WAVE];
  double v = kind == 0 ? INFINITY : -INFINITY;
  bool nan = false;
  for (int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * SDM_BLOCK) {
    const double x = a[i];
    nan |= (x != x);
    if (kind == 0) v = x < v ? x : v; else v = x > v ? x : v;
  }
  if (__any(nan) && lane_id() == 0) atomicOr(has_nan, 1);
  v = kind == 0 ? wave_min_f64(v) : wave_max_f64(v);
  if (lane_id() == 0) sm[threadIdx.x / SDM_WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < SDM_BLOCK / SDM_WAVE; ++w)
      v = kind == 0 ? (sm[w] < v ? sm[w] : v) : (sm[w] > v ? sm[w] : v);
    partial[blockIdx.x] = v;
  }
}

// one workgroup of 1024: min / max of the (at most 1024) partials (order does not matter)
__global__ void __launch_bounds__(1024)
k_reduce_final(int kind, const double *partial, int np, const int *has_nan, double *out) {
  __shared__ double sm[1024 / SDM_WAVE];
  double v = kind == 0 ? INFINITY : -INFINITY;
  if ((int)threadIdx.x < np) v = partial[threadIdx.x];
  v = kind == 0 ? wave_min_f64(v) : wave_max_f64(v);
  if (lane_id() == 0) sm[threadIdx.x / SDM_WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 1024 / SDM_WAVE; ++w)
      v = kind == 0 ? (sm[w] < v ? sm[w] : v) : (sm[w] > v ? sm[w] : v);
    out[0] = *has_nan ? NAN : v;
  }
}

extern "C" int sdm_math_eval(sdm_ctx *ctx, int fn, double *out, const double *a, const double *b,
                             int64_t n) {
  ARG_TRY(ctx && n >= 0 && fn >= SDM_MATH_EXP && fn <= SDM_MATH_LOG1P);
  ARG_TRY(n == 0 || (out && a && (b || fn != SDM_MATH_POW)));
  if (n == 0) return SDM_OK;
  hipLaunchKernelGGL(k_math_eval, dim3((unsigned)((n + SDM_BLOCK - 1) / SDM_BLOCK)), dim3(SDM_BLOCK),
                     0, ctx->stream, fn, out, a, b, n);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_reduce_f64(sdm_ctx *ctx, int kind, const double *a, int64_t n,
                              double *result) {
  ARG_TRY(ctx && a && n > 0 && result && (kind == 0 || kind == 1));
  const int nb = (int)(grid_for(n) < 1024 ? grid_for(n) : 1024);
  int rc = sdm_reserve(ctx, carve_size(sizeof(double) * 1024) + 512);
  if (rc) return rc;
  Carver cv(ctx->arena);
  double *partial = cv.take<double>(1024);
  int *has_nan = cv.take<int>(4);
  double *out = (double *)(has_nan + 2);
  HIP_TRY(hipMemsetAsync(has_nan, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_reduce_f64, dim3(nb), dim3(SDM_BLOCK), 0, ctx->stream, kind, a, n,
                     partial, has_nan);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, ctx->stream, kind, partial, nb,
                     has_nan, out);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  memcpy(result, ctx->mailbox, sizeof(double));
  return SDM_OK;
}
