// common.h -- shared host/device helpers of libsdm_hip (gfx950 only; wave = 64 lanes)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sdm_hip.h"
#include "sdm_math.h"

typedef unsigned __int128 u128;

#define SDM_BLOCK 256
#define SDM_MAX_EVENTS 8192
#define SDM_WAVE 64

#define SDM_CNT_OVERFLOW 5  // word of a counter slot that counts refused breakups (fused.hip)
#define SDM_CNT_SLOTS 128
#define SDM_CNT_STRIDE 16  // int64 per slot: one 128-B line each

struct ShufRec;  // shuffle.hip

// ---- context ---------------------------------------------------------------------------
#define SDM_DEAD_LIST_CAP 4096  // = index.hip's COMPACT_WAVES: the list is sorted in that much LDS
struct sdm_ctx {
  int device;
  hipStream_t stream;
  // scratch arena (grown on demand, never shrunk)
  char *arena;
  size_t arena_bytes;
  // PCG64 jump table: tab[b] = {A^(2^b), C_(2^b)} for the increment `tab_inc`
  u128 *pcg_tab;  // device, 64 x 2
  // jump-aheads by the distances the kernels use most, as ready affine maps {multiplier,
  // increment} (one 128-bit multiply-add instead of one per set bit of the distance): entries
  // [0, PCG_AFF_SMALL) for distances 0, 1, 2, .., then PCG_AFF_TILES entries for multiples of
  // PCG_AFF_STRIDE
  u128 *pcg_aff;
  u128 tab_inc;
  bool tab_valid;
  // pinned host mailbox for scalar read-backs
  int64_t *mailbox;  // SDM_MAILBOX_WORDS x int64, hipHostMalloc (mapped, coherent)
  // the two box slots (SDM_BOX) as the device sees them: a kernel publishes the 8 control words
  // into slot poll_seq & 1 and then the sequence number in the slot's word 8; the host polls that
  // word (9.8 us against 15.4 us for hipMemcpyAsync + hipStreamSynchronize, measured)
  int64_t *box_dev;
  int64_t poll_seq;
  // fused.hip, sdm_collision_run: time steps of a one-cell non-adaptive box replayed as a hipGraph
  // (two steps per graph: the permutation buffers are back in their roles).  The kernels of a
  // replayed step cannot get the stream positions as arguments: they read them from `gwords`
  // ({doubles drawn from the collision stream, from the breakup streams}), advanced on the device
  // at the end of every sub-step.
  uint64_t *gwords;        // device, 4 words
  void *own_stream, *own_event;  // hipStream_t / hipEvent_t of the replays
  uint64_t graph_calls;
  bool graph_capture;      // collision_step is being captured: device-side stream positions
  void *graph_exec;        // hipGraphExec_t of the cached two-step graph (NULL: none)
  void *graph_key;         // what the cached graph was captured for (memcmp'ed)
  size_t graph_key_bytes;
  int n_cus;               // fused.hip: CUs of the device (0: not asked yet)
  bool cell_attr_done;     // fused.hip: large-LDS attribute of the per-cell kernels set on this device
  int compact_grid;        // index.hip: workgroups of k_compact_persistent that are co-resident here
  int build_resident;      // index.hip: likewise k_bin_build2 (0: not asked yet, -1: unknown)
  // fused.hip: the pair kernel of the previous sub-step of this run sorted this one's events
  struct {
    bool active;
    const void *owner;
  } presorted;
  // fused.hip: which of the two sets of pair-list fill counts the previous step of the run left clean
  struct {
    bool active;
    const void *owner;
    int clean_set;
  } lists;
  int debug_box_delay_us;  // SDM_DEBUG_BOX_DELAY_US (tests): the host sleeps before each wait
  // fused.hip, cell-ordered working copy of a multi-step run: the kernels then see super-droplets
  // under call-local labels (position in the sorted permutation at the start), except
  // `normalize`'s raw look-up cell_id[pair slot] (collisions_methods.py:633-662), which keeps
  // reading the caller's column
  const int64_t *cell_id_raw;
  // fused.hip: compactions of the CURRENT call for which the closed-form re-sort is not asked for
  // (reset at every entry: which path a call takes must not depend on earlier calls)
  int resort_backoff;
  // comm.hip: RCCL communicator of sharded runs (NULL: the host's exchange callback)
  void *comm;
  bool comm_owned;
  int comm_rank, comm_world;
  int opt_resort;                // SDM_OPT_RESORT
  int64_t opt_max_substeps;      // SDM_OPT_MAX_SUBSTEPS (0: none)
  int opt_cell_shape;            // SDM_OPT_CELL_SHAPE
  int opt_records, opt_no_presort, opt_no_cell_copy;  // SDM_OPT_REC_FORMAT / _NO_PRESORT / _NO_CELL_COPY
  int64_t stats[SDM_N_STATS];    // SDM_STAT_* (host-side counters, sdm_ctx_read_stats)
  // fused.hip: what a multi-cell adaptive step knows at its end, for the next step of the same call
  // (valid length, an upper bound of the cell sizes; the state is sorted) - saves that step's
  // opening read-back
  struct {
    bool active;
    const void *owner;
    int64_t valid, max_cell, n_active_cells;
  } carry;
  // fused.hip: head of the next sub-step launched ahead of a read-back, carried over a step boundary
  struct {
    bool active;
    const void *owner;  // the sdm_step_state it belongs to
    uint64_t off_before, off_b_before;  // stream positions to return to if it is discarded
    u128 s_rand, s_rand_b;
    const void *rec, *ovf_head, *ovf_next;
    int rec_fmt;
    int64_t *cur, *alt;  // permutation buffers as its kernels left them (cur: written)
  } ahead;
  // displacement.hip (sharded step): precipitated masses by position, zero between uses
  double *rain_carry;
  int64_t rain_carry_len;
  // device control words for fine-grained calls (int64[16]; word 6: dt_left[0] of an adaptive
  // single cell for the next sub-step, fused.hip; 8: a length; 10-11: adaptive_end; 12-15: barrier)
  int64_t *dscal;
  // single-cell collision counters, spread over SDM_CNT_SLOTS cache lines (fused.hip)
  int64_t *cnt_slots;
  // adaptive steps of one cell: the positions of a sub-step's dead (SDM_DEAD_LIST_CAP words), two
  // counters on lines of their own, used in turn by sub-step number `dead_seq` (index.h:
  // CompactEpilogue)
  int64_t *dead_pos;
  unsigned long long *dead_ctr;
  uint64_t dead_seq;
  // optional per-phase timing with HIP events on the ctx stream (bench / profiling only)
  bool timing;
  hipEvent_t *ev;      // pool of SDM_MAX_EVENTS events
  int *ev_phase;       // phase id of each (begin, end) pair
  int n_ev;            // events used (2 per timed region)
  double phase_ms[SDM_N_PHASES];
  int64_t phase_count[SDM_N_PHASES];
};

// The polled copy of the control block is double-buffered: publication number `seq` goes to slot
// seq & 1 (16 words apart; words 0-7 the block, word 8 the sequence number).  One sub-step at most
// is launched ahead of the host's wait, so the publication the host waits for and the one that
// follows it never share a slot.
#define SDM_BOX 16         // first mailbox word of slot 0
#define SDM_BOX_STRIDE 16  // words between the two slots
#define SDM_MAILBOX_WORDS 64  // (words 48..63: stream positions handed to graph replays)
int sdm_wait_box(sdm_ctx *ctx, int64_t seq);  // ctx.hip
// waits for publication `seq` and copies its eight words; fails if the slot was overwritten while
// it was read (cannot happen with one publication in flight ahead: checked, not assumed)
int sdm_read_box(sdm_ctx *ctx, int64_t seq, int64_t out[8]);

// control word 7: low byte = device-side error code, bit 8 = event "a cell's stats_dt_min became
// equal to dt_min" (the host then evaluates the reference's condition, collision.py:276-277)
#define SDM_CTL7_ERROR_MASK 0xff
#define SDM_CTL7_DT_MIN 0x100

#ifdef __HIPCC__
// (k_cells_adaptive: any thread of the launch may raise the bit; the multi-cell per-cell route has
// no use for it any more - k_cells_turn's workgroup 0 derives the event from the minima it reads
// anyway and writes the word itself, so that the publication it makes cannot overtake it.  The
// atomic is a returning one with its result consumed: complete at the memory side before the wave
// goes on - round 3's fence-free finish ticket needed that, and it costs nothing to keep)
__device__ __forceinline__ void note_dt_min(int64_t *ctl, double stats_value, double dt_min) {
  if (stats_value == dt_min) {
    const unsigned long long was =
        atomicOr((unsigned long long *)&ctl[7], (unsigned long long)SDM_CTL7_DT_MIN);
    asm volatile("" ::"v"(was));
  }
}

// last act of a one-thread epilogue: control block -> host-visible box, then the sequence number
// (work: what the host is to see as working length, word 1)
// Every word carries the low 24 bits of the sequence number above its value (40 bits: lengths,
// flags, counts): the host accepts a block only when all eight words carry the tag it waits for.
// The nine stores travel to host memory one by one, and nothing makes them ARRIVE in order (found
// by tests/fuzz_sharded_flow.py with two processes on one card: the sequence word of a publication
// was seen before its data words, the host acted on the block of two publications before - a death
// went unnoticed and the step ended with a flagged super-droplet in the permutation)
#define SDM_BOX_TAG_SHIFT 40
#define SDM_BOX_VALUE_MASK ((1ull << SDM_BOX_TAG_SHIFT) - 1)
#define SDM_BOX_TAG_MASK 0xFFFFFFull
__device__ __forceinline__ void publish_ctl(const int64_t *ctl, int64_t *box, int64_t seq,
                                            int64_t work) {
  if (!box) return;
  box += (seq & 1) * SDM_BOX_STRIDE;
  const uint64_t tag = ((uint64_t)seq & SDM_BOX_TAG_MASK) << SDM_BOX_TAG_SHIFT;
  for (int w = 0; w < 8; ++w) {
    const int64_t v = w == 1 ? work
                             : __hip_atomic_load(&ctl[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&box[w], (int64_t)(tag | ((uint64_t)v & SDM_BOX_VALUE_MASK)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __hip_atomic_store(&box[8], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

void sdm_phase_begin(sdm_ctx *ctx, int phase);
void sdm_phase_end(sdm_ctx *ctx);
struct PhaseScope {
  sdm_ctx *c;
  PhaseScope(sdm_ctx *ctx, int phase) : c(ctx) { if (c->timing) sdm_phase_begin(c, phase); }
  ~PhaseScope() { if (c->timing) sdm_phase_end(c); }
};

void sdm_set_error(const char *fmt, ...);
// comm.hip: one exchange of a sharded step - RCCL if the context has a communicator, else the callback
int sdm_exchange(sdm_ctx *ctx, sdm_exchange_fn callback, void *user, int what, void *buffer,
                 int64_t count);
int sdm_reserve(sdm_ctx *ctx, size_t bytes);

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      sdm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,  \
                    __LINE__);                                                         \
      return SDM_E_HIP;                                                                \
    }                                                                                  \
  } while (0)

#define ARG_TRY(cond)                                                          \
  do {                                                                         \
    if (!(cond)) {                                                             \
      sdm_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);    \
      return SDM_E_ARG;                                                        \
    }                                                                          \
  } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

static inline unsigned grid_for(int64_t n, int per_block = SDM_BLOCK) {
  int64_t g = (n + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : g);
}

// carve helper for the scratch arena (256-B aligned pieces)
struct Carver {
  char *base;
  size_t off;
  explicit Carver(char *b) : base(b), off(0) {}
  template <typename T>
  T *take(size_t n) {
    T *p = (T *)(base + off);
    off += ((n * sizeof(T) + 255) / 256) * 256;
    return p;
  }
};
static inline size_t carve_size(size_t bytes) { return ((bytes + 255) / 256) * 256; }

// ---- device helpers --------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// largest c in [0, n_cell) with cell_start[c] <= i  (needs cell_start[0] <= i < cell_start[n_cell])
__device__ __forceinline__ int64_t find_cell(const int64_t *__restrict__ cs, int64_t n_cell,
                                             int64_t i) {
  int64_t lo = 0, hi = n_cell;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (cs[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- PCG64 (NumPy): XSL-RR 128/64 ----
#define PCG_MULT_HI 0x2360ED051FC65DA4ULL
#define PCG_MULT_LO 0x4385DF649FCCF645ULL

__host__ __device__ __forceinline__ u128 pcg_mult() {
  return (((u128)PCG_MULT_HI) << 64) | PCG_MULT_LO;
}

__host__ __device__ __forceinline__ double pcg_output(u128 state) {
  const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
  const uint64_t x = hi ^ lo;
  const unsigned r = (unsigned)(hi >> 58);
  const uint64_t v = (x >> r) | (x << ((64 - r) & 63));
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}

#define PCG_AFF_SMALL 4097
#define PCG_AFF_TILES 8192
#define PCG_AFF_STRIDE 4096
// state after the jump of entry k of ctx->pcg_aff
__device__ __forceinline__ u128 pcg_apply(u128 state, const u128 *__restrict__ aff, int64_t k) {
  return state * aff[2 * k] + aff[2 * k + 1];
}

// state after `delta` further draws, using the per-increment jump table
__device__ __forceinline__ u128 pcg_jump(u128 state, const u128 *__restrict__ tab,
                                         uint64_t delta) {
  int b = 0;
  while (delta) {
    if (delta & 1) state = state * tab[2 * b] + tab[2 * b + 1];
    delta >>= 1;
    ++b;
  }
  return state;
}

// the same through the ready affine maps when the distance is within their reach (two multiply-adds
// instead of one per set bit, and no dependent table reads)
__device__ __forceinline__ u128 pcg_jump_fast(u128 state, const u128 *__restrict__ tab,
                                              const u128 *__restrict__ aff, uint64_t delta) {
  if (aff && delta < (uint64_t)PCG_AFF_TILES * PCG_AFF_STRIDE)
    return pcg_apply(pcg_apply(state, aff, PCG_AFF_SMALL + (int64_t)(delta / PCG_AFF_STRIDE)), aff,
                     (int64_t)(delta % PCG_AFF_STRIDE));
  return pcg_jump(state, tab, delta);
}

// wave-level reductions (all 64 lanes must participate)
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((long long)v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// counter[cid] += v for the lanes with `active`: one atomic per wave when they all share cid
// (same-address atomics serialise in L2).  Wave-collective: every lane of the wave calls it.
__device__ __forceinline__ void wave_counter_add(int64_t *__restrict__ counter, int64_t cid,
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

// atomic min on non-negative doubles through their (order-preserving) bit pattern
__device__ __forceinline__ void atomic_min_pos_f64(double *addr, double v) {
  atomicMin((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// Python-style round-half-even to int64 (round() in collisions_methods.py:124-125)
__device__ __forceinline__ int64_t py_round(double x) { return (int64_t)rint(x); }

#endif  // __HIPCC__
