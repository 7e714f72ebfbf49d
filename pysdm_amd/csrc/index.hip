// index.hip -- permutation-side kernels: identity, the parallel-equivalent of the reference's
// serial swap chains (shuffle_local / shuffle_global), cell sort, compaction.
#include "common.h"
#include "index.h"
#include "shuffle_device.h"
#include "shuffle_build.h"

// ---------------------------------------------------------------------------------------
// identity_index  (index_methods.py:14-20)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK) k_identity(int64_t *idx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i < n) idx[i] = i;
}

extern "C" int sdm_identity_index(sdm_ctx *ctx, int64_t *idx, int64_t n) {
  ARG_TRY(ctx && n >= 0 && (idx || n == 0));
  if (n == 0) return SDM_OK;
  hipLaunchKernelGGL(k_identity, dim3(grid_for(n)), dim3(SDM_BLOCK), 0, ctx->stream, idx, n);
  LAUNCH_CHECK();
  return SDM_OK;
}

// ---------------------------------------------------------------------------------------
// Shuffle.  The reference (index_methods.py:22-43) runs, per cell [lo, hi), the serial chain
//     for i = hi-1 .. lo+1:  j_i = target(i);  swap(idx[i], idx[j_i])
// (local: j_i = int(lo + u01[i]*(hi-lo)) anywhere in the cell; global: j_i = int(u01[i]*(i+1))).
// All (i, j_i) are known up front, so the final content of position p is found by walking the
// swap history backwards: the last event touching p is the one with the SMALLEST index among
// {p itself (its "own" event)} U {i : j_i == p}; the content then came from the event's other
// end, at the moment just before that event, i.e. after all events with a LARGER index; repeat
// until a position has no earlier event, whose initial content is the answer.  Expected walk
// length is ~2 events per position.  Build: per-position record {own target, #hits, first two
// hitting events inline, rest in an overflow list}.  Bit-identical to the serial chain.
// ---------------------------------------------------------------------------------------
struct __align__(16) ShufRec {
  int32_t j;    // target of this position's own event, -1 if none (first slot of its cell)
  int32_t cnt;  // number of events i with j_i == this position
  int32_t s0, s1;  // the first two of them in arrival order (any order; all get scanned)
};

#define SHUF_ELEMS 4  // positions per thread in the build kernel

// RNG = true: u01[i] is draw number (offset + i) of the PCG64 stream, generated in place
// (`s_off` = generator state after `offset` draws); RNG = false: u01 read from memory.
template <bool GLOBAL, bool RNG>
__global__ void __launch_bounds__(SDM_BLOCK)
k_shuffle_build(ShufRec *__restrict__ rec, int32_t *__restrict__ ovf_head,
                int32_t *__restrict__ ovf_next, const double *__restrict__ u01,
                const int64_t *__restrict__ cell_start, int64_t n_cell,
                const int64_t *__restrict__ p_length, int64_t length_arg, u128 s_off, u128 inc,
                const u128 *__restrict__ tab) {
  const int64_t length = p_length ? *p_length : length_arg;
  const int64_t blk_first = (int64_t)blockIdx.x * (SDM_BLOCK * SHUF_ELEMS);
  if (blk_first >= length) return;
  const int64_t first = blk_first + (int64_t)threadIdx.x * SHUF_ELEMS;
  double u[SHUF_ELEMS];
  if (RNG) {
    __shared__ u128 s_blk;
    if (threadIdx.x == 0) s_blk = pcg_jump(s_off, tab, (uint64_t)blk_first);
    __syncthreads();
    u128 state = pcg_jump(s_blk, tab, (uint64_t)threadIdx.x * SHUF_ELEMS);
    const u128 mult = pcg_mult();
#pragma unroll
    for (int e = 0; e < SHUF_ELEMS; ++e) {
      state = state * mult + inc;
      u[e] = pcg_output(state);
    }
  } else {
#pragma unroll
    for (int e = 0; e < SHUF_ELEMS; ++e) u[e] = first + e < length ? u01[first + e] : 0.0;
  }
  int64_t j[SHUF_ELEMS];
  int64_t lo = 0, hi = 0;
  bool have_cell = false;
#pragma unroll
  for (int e = 0; e < SHUF_ELEMS; ++e) {
    const int64_t i = first + e;
    j[e] = -1;
    if (i >= length) continue;
    if (GLOBAL) {
      if (i >= 1) {
        const int64_t t = (int64_t)(u[e] * (double)(i + 1));
        j[e] = t > i ? i : t;
      }
    } else {
      if (!have_cell || i >= hi) {
        const int64_t c = n_cell == 1 ? 0 : find_cell(cell_start, n_cell, i);
        lo = cell_start[c];
        hi = cell_start[c + 1];
        have_cell = true;
      }
      if (i > lo) {
        const int64_t t = (int64_t)((double)lo + u[e] * (double)(hi - lo));
        // memory safety only: the reference would index past the cell with prob ~2^-43
        j[e] = t > hi - 1 ? hi - 1 : (t < lo ? lo : t);
      }
    }
  }
  // own-event targets: 4 consecutive records, stored as one row of int32 j fields
  int slot[SHUF_ELEMS];
#pragma unroll
  for (int e = 0; e < SHUF_ELEMS; ++e) {
    if (first + e < length) rec[first + e].j = (int32_t)j[e];
    slot[e] = j[e] >= 0 ? atomicAdd(&rec[j[e]].cnt, 1) : -1;  // independent: all in flight
  }
#pragma unroll
  for (int e = 0; e < SHUF_ELEMS; ++e) {
    if (slot[e] < 0) continue;
    const int32_t i = (int32_t)(first + e);
    if (slot[e] == 0) rec[j[e]].s0 = i;
    else if (slot[e] == 1) rec[j[e]].s1 = i;
    else ovf_next[i] = atomicExch(&ovf_head[j[e]], i);
  }
}

// positions [length, n_total) keep their content (copied through)
template <bool GLOBAL>
__global__ void __launch_bounds__(SDM_BLOCK)
k_shuffle_trace(int64_t *__restrict__ out, const int64_t *__restrict__ idx0,
                const ShufRec *__restrict__ rec, const int32_t *__restrict__ ovf_head,
                const int32_t *__restrict__ ovf_next, const int64_t *__restrict__ cell_start,
                int64_t n_cell, const int64_t *__restrict__ p_length, int64_t length_arg,
                int64_t n_total) {
  const int64_t length = p_length ? *p_length : length_arg;
  const int64_t p = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (p >= length) {
    if (p < n_total) out[p] = idx0[p];
    return;
  }
  int32_t e;  // only events with index > e are still "in the past" of the walk
  if (GLOBAL) {
    e = 0;
  } else {
    const int64_t c = n_cell == 1 ? 0 : find_cell(cell_start, n_cell, p);
    e = (int32_t)cell_start[c];
  }
  int32_t q = (int32_t)p;
  for (;;) {
    const ShufRec r = rec[q];
    int32_t best = INT32_MAX;
    if (q > e && r.j >= 0) best = q;
    if (r.cnt > 0 && r.s0 > e && r.s0 < best) best = r.s0;
    if (r.cnt > 1 && r.s1 > e && r.s1 < best) best = r.s1;
    if (r.cnt > 2) {
      // overflow list: exactly cnt-2 nodes (the list head needs no initialisation)
      int32_t t = ovf_head[q];
      for (int c = 2; c < r.cnt; ++c) {
        if (t > e && t < best) best = t;
        t = ovf_next[t];
      }
    }
    if (best == INT32_MAX) break;
    // event `best` exchanged positions (best, j_best); q is one end, continue at the other
    q = (best == q) ? r.j : best;
    e = best;
  }
  out[p] = idx0[q];
}

static size_t shuffle_scratch_bytes(int64_t n) {
  return carve_size(sizeof(ShufRec) * n) + 2 * carve_size(sizeof(int32_t) * n) +
         carve_size(sizeof(int64_t) * n);
}

static bool binned_ok(int64_t n, bool global);
static int shuffle_binned_async(sdm_ctx *ctx, char *scratch, int64_t *out, const int64_t *idx0,
                                const double *u01, const int64_t *cell_start, int64_t n_cell,
                                const int64_t *p_length, int64_t length_bound, int64_t n_total,
                                u128 s_off, u128 inc, ShuffleViews *views, int64_t id_bound = -1,
                                const uint64_t *dev_off = nullptr,
                                const SortPrologue *presorted = nullptr);

// out-of-place core: out[0:length) = shuffled idx0[0:length), out[length:n_total) = idx0[...].
// u01 == nullptr: draws generated in the kernel from (rng_state_inc, rng_offset).
int sdm_shuffle_async(sdm_ctx *ctx, char *scratch, int64_t *out, const int64_t *idx0,
                      const double *u01, const int64_t *cell_start, int64_t n_cell,
                      const int64_t *p_length, int64_t length_bound, bool global,
                      int64_t n_total, const uint64_t *rng_state_inc, uint64_t rng_offset) {
  if (length_bound <= 0) return SDM_OK;
  Carver cv(scratch);
  ShufRec *rec = cv.take<ShufRec>(length_bound);
  int32_t *ovf_head = cv.take<int32_t>(length_bound);
  int32_t *ovf_next = cv.take<int32_t>(length_bound);
  u128 s_off = 0, inc = 0;
  if (!u01) {
    int rc = sdm_pcg_prepare(ctx, rng_state_inc);
    if (rc) return rc;
    const u128 st = (((u128)rng_state_inc[0]) << 64) | rng_state_inc[1];
    inc = (((u128)rng_state_inc[2]) << 64) | rng_state_inc[3];
    s_off = sdm_pcg_advance_host(st, inc, rng_offset);
  }
  if (binned_ok(length_bound, global))
    return shuffle_binned_async(ctx, scratch, out, idx0, u01, cell_start, n_cell, p_length,
                                length_bound, n_total, s_off, inc, nullptr);
  {
    PhaseScope ph(ctx, SDM_PHASE_SHUFFLE_CLEAR);
    HIP_TRY(hipMemsetAsync(rec, 0, sizeof(ShufRec) * length_bound, ctx->stream));
  }
  const dim3 block(SDM_BLOCK);
  {
    PhaseScope ph(ctx, SDM_PHASE_SHUFFLE_BUILD);
    const dim3 grid(grid_for(length_bound, SDM_BLOCK * SHUF_ELEMS));
#define LAUNCH_BUILD(G, R)                                                                    \
  hipLaunchKernelGGL((k_shuffle_build<G, R>), grid, block, 0, ctx->stream, rec, ovf_head,     \
                     ovf_next, u01, cell_start, n_cell, p_length, length_bound, s_off, inc,   \
                     ctx->pcg_tab)
    if (global) { if (u01) LAUNCH_BUILD(true, false); else LAUNCH_BUILD(true, true); }
    else { if (u01) LAUNCH_BUILD(false, false); else LAUNCH_BUILD(false, true); }
#undef LAUNCH_BUILD
    LAUNCH_CHECK();
  }
  {
    PhaseScope ph(ctx, SDM_PHASE_SHUFFLE_TRACE);
    const int64_t span = n_total > length_bound ? n_total : length_bound;
    const dim3 grid(grid_for(span));
    if (global)
      hipLaunchKernelGGL(k_shuffle_trace<true>, grid, block, 0, ctx->stream, out, idx0, rec,
                         ovf_head, ovf_next, cell_start, n_cell, p_length, length_bound, n_total);
    else
      hipLaunchKernelGGL(k_shuffle_trace<false>, grid, block, 0, ctx->stream, out, idx0, rec,
                         ovf_head, ovf_next, cell_start, n_cell, p_length, length_bound, n_total);
    LAUNCH_CHECK();
  }
  return SDM_OK;
}

// ---------------------------------------------------------------------------------------
// Binned build (local croupier): the same event records without a single global atomic.
// Positions are cut into bins of BIN_POS; an event (i, j_i) has to reach the bin of its target
// j_i: every event tile orders its events by target bin in LDS and writes them back tile-major
// (k_bin_sort), then one workgroup per bin gathers its runs from all tiles, assembles the records
// of its BIN_POS positions in LDS and writes them out whole (k_bin_build2):
//   PackRec {own target j (-1 none), first hit s0 (-1 none), second hit s1 (-1 none),
//            initial content of the position (int32) | bit 31 = "more hits in the overflow list"}
// so the backward walk needs no separate gather of the initial content.  u01 comes either from
// memory or from the PCG64 stream evaluated in place (16 consecutive draws per thread).
// ---------------------------------------------------------------------------------------
// (tile shapes, targets_run, block_excl_scan, bin_sort_body: shuffle_build.h)
#ifdef BIN_PROFILE
__device__ long long bin_prof[32];
extern "C" int sdm_debug_bin_profile(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(bin_prof), sizeof(long long) * 32) == hipSuccess ? 0 : -2;
}
#endif

template <bool RNG, int TILE>
__global__ void k_bin_sort(int2 *events, int32_t *toff, int32_t *jarr, int32_t *loc, int n_bins,
                           const double *u01, const int64_t *cell_start, int64_t n_cell,
                           const int64_t *p_length, int64_t length_arg, u128 s_off, u128 inc,
                           const u128 *tab, const uint64_t *dev_off, const u128 *aff);

// (BuildPrologue: index.h; the kernel is defined after the compaction code it may have to run)
template <int FMT>
__global__ void k_bin_build2(void *rec_out, int32_t *ovf_head, int32_t *ovf_next,
                             const int2 *events, const int32_t *toff, const int32_t *jarr,
                             int n_bins, int n_tiles, int ev_tile, const int64_t *idx0,
                             const int64_t *p_length, int64_t length_arg, BuildPrologue P);

// backward walk over packed records; positions [length, n_total) are copied through
__global__ void __launch_bounds__(SDM_BLOCK)
k_trace_packed(int64_t *__restrict__ out, const int64_t *__restrict__ idx0,
               const PackRec *__restrict__ rec, const int32_t *__restrict__ ovf_head,
               const int32_t *__restrict__ ovf_next, const int64_t *__restrict__ cell_start,
               int64_t n_cell, const int64_t *__restrict__ p_length, int64_t length_arg,
               int64_t n_total) {
  const int64_t length = p_length ? *p_length : length_arg;
  const int64_t p = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (p >= length) {
    if (p < n_total) out[p] = idx0[p];
    return;
  }
  const int64_t c = n_cell == 1 ? 0 : find_cell(cell_start, n_cell, p);
  out[p] = rec_id(walk_packed(rec, ovf_head, ovf_next, (int32_t)p, (int32_t)cell_start[c]));
}

static int bin_count(int64_t n) { return (int)((n + BIN_POS - 1) / BIN_POS); }
static int ev_tile_count(int64_t n) { return (int)((n + EV_TILE - 1) / EV_TILE); }

static size_t binned_scratch_bytes(int64_t n) {
  const size_t nb = (size_t)bin_count(n), nt = (size_t)ev_tile_count(n);
  // (either tile size: the larger tile pads more, the smaller has more rows of offsets)
  return carve_size(sizeof(PackRec) * (n + EV_TILE_BIG)) + 3 * carve_size(sizeof(int32_t) * n) +
         carve_size(sizeof(int2) * (n + EV_TILE_BIG)) + carve_size(sizeof(int32_t) * (nb + 1) * nt);
}

// usable while the count matrix stays small and LDS holds the per-bin arrays of K3
// (8192 bins = 2^25 positions: K3 then holds 3 x 8193 + 3 x 4096 ints = 146 KB of the CU's 160 KB)
static bool binned_ok(int64_t n, bool global) { return !global && n >= 2 && bin_count(n) <= 8192; }

// SDM_OPT_REC_FORMAT = records: the round-1..3 records also where the successor words would be built
static bool chain_enabled(const sdm_ctx *ctx) { return !ctx->opt_records; }

static int shuffle_binned_async(sdm_ctx *ctx, char *scratch, int64_t *out, const int64_t *idx0,
                                const double *u01, const int64_t *cell_start, int64_t n_cell,
                                const int64_t *p_length, int64_t length_bound, int64_t n_total,
                                u128 s_off, u128 inc, ShuffleViews *views, int64_t id_bound,
                                const uint64_t *dev_off, const SortPrologue *presorted) {
  // id_bound: the ids in idx0 are below it (-1: unknown); decides the record layout
  const int64_t both = id_bound > length_bound ? id_bound : length_bound;
  // SDM_REC_CHAIN where the caller's kernels do the walk (`views`); SDM_REC_FORMAT=records keeps
  // the round-1..3 records there as well (A/B measurements)
  // ... and only while a bin's run in a tile is a whole 64-byte sector of S words (16 events: at
  // most 256 bins = 2^20 positions).  Beyond, the S words go back as quarter sectors written by
  // different workgroups and the build loses more than the walk gains (2^22: k_bin_build2 62 ->
  // 157 us against k_pair_prob 258 -> 195; profiles/r04_chain_at_2p22.json)
  // From there to 2^22 (1024 bins) tiles of 16384 events restore the whole sectors: CHAIN with
  // them took straub (2^22, adaptive, breakup) from 3.95 to 4.20e9 pairs/s, straub_rain from 4.42
  // to 4.73e9 (profiles/r04_tile16k.json).  Only for these builds: the 139 KB of LDS of such a
  // tile sort leave one workgroup per CU, which the 4096-event sort of the other routes need not pay
  const bool chain_ok = views && id_bound >= 0 && both <= CHAIN_MAX && chain_enabled(ctx);
  // (nor where the tile sort can ride in the previous pair kernel: that one is 4096 events)
  const int tile = (chain_ok && !presorted && bin_count(length_bound) * 16 > EV_TILE &&
                    bin_count(length_bound) * 16 <= EV_TILE_BIG &&
                    !sdm_shuffle_presort_ok(ctx, length_bound, id_bound)) ? EV_TILE_BIG : EV_TILE;
  const bool whole_sectors = bin_count(length_bound) * 16 <= tile;
  const int fmt = (chain_ok && whole_sectors)
                      ? SDM_REC_CHAIN
                  : id_bound < 0 ? SDM_REC_PLAIN
                  : both <= P21_MAX ? SDM_REC_P21 : (both <= P24_MAX ? SDM_REC_P24 : SDM_REC_PLAIN);
  const int slots = fmt == SDM_REC_CHAIN ? 5 : fmt == SDM_REC_P21 ? 4 : (fmt == SDM_REC_P24 ? 3 : 2);
  Carver cv(scratch);
  const int nb = bin_count(length_bound), nt = (int)((length_bound + tile - 1) / tile);
  PackRec *rec = cv.take<PackRec>(length_bound + tile);
  int32_t *ovf_head = cv.take<int32_t>(length_bound);
  int32_t *ovf_next = cv.take<int32_t>(length_bound);
  int32_t *jarr = cv.take<int32_t>(length_bound);
  int2 *events = cv.take<int2>((size_t)nt * tile);
  int32_t *toff = cv.take<int32_t>((size_t)(nb + 1) * nt);
  // SDM_REC_CHAIN: the record buffer's 16 B per position hold four int32 arrays of nt * EV_TILE
  // words instead - first | overflow links | loc (event -> place in `events`) | ssucc (by place)
  const size_t padded = (size_t)nt * tile;
  int32_t *chain = (int32_t *)rec;
  int32_t *loc = fmt == SDM_REC_CHAIN ? chain + 2 * padded : nullptr;
  const dim3 block(BIN_THREADS);
  const size_t lds_sort = sizeof(int32_t) * (size_t)((nb + 1) + ((nb + 1) & ~1) + 2) +
                          sizeof(int2) * tile;
  // (64 KB with three inline slots - two workgroups per CU - 80 KB with four; the prologue of a
  // presorted build sorts in the same memory)
  // (SDM_REC_CHAIN: + own target, id and place of every position of the bin)
  size_t lds_build = sizeof(int32_t) * (size_t)((slots + 1 + (fmt == SDM_REC_CHAIN ? 3 : 0)) * BIN_POS);
  if (presorted && lds_sort > lds_build) lds_build = lds_sort;
  // gfx950 has 160 KiB of LDS per CU; > 64 KiB dynamic needs opting in (per kernel and device)
  if (lds_sort > 65536) {
    HIP_TRY(hipFuncSetAttribute(tile == EV_TILE_BIG ? (const void *)k_bin_sort<true, EV_TILE_BIG>
                                                    : (const void *)k_bin_sort<true, EV_TILE>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
    HIP_TRY(hipFuncSetAttribute((const void *)k_bin_sort<false, EV_TILE>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
  }
  if (lds_build > 65536)
    HIP_TRY(hipFuncSetAttribute(fmt == SDM_REC_CHAIN ? (const void *)k_bin_build2<SDM_REC_CHAIN>
                                : fmt == SDM_REC_P21 ? (const void *)k_bin_build2<SDM_REC_P21>
                                : fmt == SDM_REC_P24 ? (const void *)k_bin_build2<SDM_REC_P24>
                                                     : (const void *)k_bin_build2<SDM_REC_PLAIN>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_build));
  {
    PhaseScope ph(ctx, SDM_PHASE_SHUFFLE_BUILD);
    BuildPrologue bp;
    memset(&bp, 0, sizeof(bp));
    if (presorted) {  // sorted by the previous pair kernel; what the build needs to redo it
      bp.compact = *presorted;
      bp.events = events;
      bp.toff = toff;
      bp.jarr = jarr;
      bp.loc = loc;
      bp.s_off = s_off;
      bp.inc = inc;
      bp.tab = ctx->pcg_tab;
      bp.aff = ctx->pcg_aff;
    } else
    if (u01)  // (never with the large tile: `views` come with the in-kernel generator)
      hipLaunchKernelGGL((k_bin_sort<false, EV_TILE>), dim3(nt), block, lds_sort, ctx->stream,
                         events, toff, jarr, loc, nb, u01, cell_start, n_cell, p_length,
                         length_bound, s_off, inc, ctx->pcg_tab, (const uint64_t *)nullptr,
                         (const u128 *)nullptr);
    else if (tile == EV_TILE_BIG)
      hipLaunchKernelGGL((k_bin_sort<true, EV_TILE_BIG>), dim3(nt), block, lds_sort, ctx->stream,
                         events, toff, jarr, loc, nb, u01, cell_start, n_cell, p_length,
                         length_bound, s_off, inc, ctx->pcg_tab, dev_off,
                         (const u128 *)ctx->pcg_aff);
    else
      hipLaunchKernelGGL((k_bin_sort<true, EV_TILE>), dim3(nt), block, lds_sort, ctx->stream,
                         events, toff, jarr, loc, nb, u01, cell_start, n_cell, p_length,
                         length_bound, s_off, inc, ctx->pcg_tab, dev_off,
                         (const u128 *)ctx->pcg_aff);
#define BUILD_LAUNCH(F)                                                                        \
  hipLaunchKernelGGL(k_bin_build2<F>, dim3(nb), block, lds_build, ctx->stream, (void *)rec,    \
                     ovf_head, ovf_next, events, toff, jarr, nb, nt, tile, idx0, p_length,          \
                     length_bound, bp)
    bp.chain_links = chain + padded;
    bp.loc = loc;
    bp.ssucc = (uint32_t *)(chain + 3 * padded);
    if (fmt == SDM_REC_CHAIN) BUILD_LAUNCH(SDM_REC_CHAIN);
    else if (fmt == SDM_REC_P21) BUILD_LAUNCH(SDM_REC_P21);
    else if (fmt == SDM_REC_P24) BUILD_LAUNCH(SDM_REC_P24);
    else BUILD_LAUNCH(SDM_REC_PLAIN);
#undef BUILD_LAUNCH
    LAUNCH_CHECK();
  }
  if (views) {  // build only: the caller's kernels do the walk
    views->rec = rec;
    views->fmt = fmt;
    views->ovf_head = ovf_head;
    // (SDM_REC_CHAIN: first = rec, tsucc = ovf_head, ssucc by place in the sorted array)
    views->ovf_next = fmt == SDM_REC_CHAIN ? chain + 3 * padded : ovf_next;
    return SDM_OK;
  }
  {
    PhaseScope ph(ctx, SDM_PHASE_SHUFFLE_TRACE);
    const int64_t span = n_total > length_bound ? n_total : length_bound;
    hipLaunchKernelGGL(k_trace_packed, dim3(grid_for(span)), dim3(SDM_BLOCK), 0, ctx->stream, out,
                       idx0, rec, ovf_head, ovf_next, cell_start, n_cell, p_length, length_bound,
                       n_total);
    LAUNCH_CHECK();
  }
  return SDM_OK;
}

// records only (binned build, local croupier, in-kernel PCG64): the walk is left to the caller
bool sdm_shuffle_can_split(int64_t n, bool global) { return binned_ok(n, global); }

int sdm_shuffle_build_async(sdm_ctx *ctx, char *scratch, const int64_t *idx0,
                            const int64_t *cell_start, int64_t n_cell, const int64_t *p_length,
                            int64_t length_bound, const uint64_t *rng_state_inc,
                            uint64_t rng_offset, ShuffleViews *views, int64_t id_bound,
                            const uint64_t *dev_off, const SortPrologue *presorted) {
  int rc = sdm_pcg_prepare(ctx, rng_state_inc);
  if (rc) return rc;
  const u128 st = (((u128)rng_state_inc[0]) << 64) | rng_state_inc[1];
  const u128 inc = (((u128)rng_state_inc[2]) << 64) | rng_state_inc[3];
  // dev_off: the kernels add the stream position themselves (graph replay)
  const u128 s_off = dev_off ? st : sdm_pcg_advance_host(st, inc, rng_offset);
  return shuffle_binned_async(ctx, scratch, nullptr, idx0, nullptr, cell_start, n_cell, p_length,
                              length_bound, 0, s_off, inc, views, id_bound, dev_off, presorted);
}

void sdm_shuffle_sort_buffers(sdm_ctx *ctx, char *scratch, int64_t length_bound, SortBuffers *out) {
  Carver cv(scratch);  // (the carve of shuffle_binned_async)
  const int nb = bin_count(length_bound), nt = ev_tile_count(length_bound);
  int32_t *chain = (int32_t *)cv.take<PackRec>(length_bound + EV_TILE);
  (void)cv.take<int32_t>(length_bound);
  (void)cv.take<int32_t>(length_bound);
  out->loc = chain_enabled(ctx) && length_bound <= CHAIN_MAX && nb * 16 <= EV_TILE
                 ? chain + 2 * (size_t)nt * EV_TILE : nullptr;
  out->jarr = cv.take<int32_t>(length_bound);
  out->events = cv.take<int2>((size_t)nt * EV_TILE);
  out->toff = cv.take<int32_t>((size_t)(nb + 1) * nt);
  out->n_bins = nb;
  out->n_tiles = nt;
  out->lds_bytes = sizeof(int32_t) * (size_t)((nb + 1) + ((nb + 1) & ~1) + 2) +
                   sizeof(int2) * EV_TILE;
}

size_t sdm_shuffle_scratch(int64_t n) {
  const size_t a = shuffle_scratch_bytes(n) - carve_size(sizeof(int64_t) * n);
  const size_t b = binned_scratch_bytes(n);
  return a > b ? a : b;
}

extern "C" int sdm_shuffle_global(sdm_ctx *ctx, int64_t *idx, int64_t length,
                                  const double *u01) {
  ARG_TRY(ctx && length >= 0 && length < INT32_MAX && (length == 0 || (idx && u01)));
  if (length < 2) return SDM_OK;
  int rc = sdm_reserve(ctx, sdm_shuffle_scratch(length) + carve_size(sizeof(int64_t) * length));
  if (rc) return rc;
  int64_t *out = (int64_t *)(ctx->arena + sdm_shuffle_scratch(length));
  rc = sdm_shuffle_async(ctx, ctx->arena, out, idx, u01, nullptr, 1, nullptr, length, true, 0,
                         nullptr, 0);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(idx, out, sizeof(int64_t) * length, hipMemcpyDeviceToDevice,
                         ctx->stream));
  return SDM_OK;
}

// shuffle_local needs the working length = cell_start[n_cell], which lives on the device: the
// kernels read it from there; the launch is sized by `length_bound` (<= len(idx)).
extern "C" int sdm_shuffle_local(sdm_ctx *ctx, int64_t *idx, const double *u01,
                                 const int64_t *cell_start, int64_t n_cell) {
  ARG_TRY(ctx && idx && u01 && cell_start && n_cell >= 1);
  // one small read-back keeps this fine-grained entry simple (the fused path avoids it)
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, cell_start + n_cell, sizeof(int64_t),
                         hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int64_t length = ctx->mailbox[0];
  ARG_TRY(length >= 0 && length < INT32_MAX);
  if (length < 2) return SDM_OK;
  int rc = sdm_reserve(ctx, sdm_shuffle_scratch(length) + carve_size(sizeof(int64_t) * length));
  if (rc) return rc;
  int64_t *out = (int64_t *)(ctx->arena + sdm_shuffle_scratch(length));
  rc = sdm_shuffle_async(ctx, ctx->arena, out, idx, u01, cell_start, n_cell, nullptr, length,
                         false, 0, nullptr, 0);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(idx, out, sizeof(int64_t) * length, hipMemcpyDeviceToDevice,
                         ctx->stream));
  return SDM_OK;
}

// ---------------------------------------------------------------------------------------
// sort_by_key (index_methods.py:46-48): idx[:] = argsort(keys, kind="stable")[::-1]
// n = number of cells; O(n^2 / threads) rank counting, keys staged through LDS tiles.
// ---------------------------------------------------------------------------------------
// one workgroup per key i: rank(i) = #{j : key_j < key_i or (key_j == key_i and j < i)}
__global__ void __launch_bounds__(SDM_BLOCK)
k_sort_by_key(int64_t *__restrict__ idx, const double *__restrict__ keys, int64_t n) {
  __shared__ int sm[SDM_BLOCK / SDM_WAVE];
  const int64_t i = blockIdx.x;
  const double ki = keys[i];
  int rank = 0;
  for (int64_t j = threadIdx.x; j < n; j += SDM_BLOCK) {
    const double kj = keys[j];
    rank += (kj < ki) || (kj == ki && j < i);
  }
  rank = wave_sum_i32(rank);
  if (lane_id() == 0) sm[threadIdx.x / SDM_WAVE] = rank;
  __syncthreads();
  if (threadIdx.x == 0) idx[n - 1 - (sm[0] + sm[1] + sm[2] + sm[3])] = i;
}

int sdm_sort_by_key_async(sdm_ctx *ctx, int64_t *idx, const double *keys, int64_t n) {
  if (n <= 0) return SDM_OK;
  hipLaunchKernelGGL(k_sort_by_key, dim3((unsigned)n), dim3(SDM_BLOCK), 0, ctx->stream, idx,
                     keys, n);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_sort_by_key(sdm_ctx *ctx, int64_t *idx, const double *keys, int64_t n) {
  ARG_TRY(ctx && n >= 0 && (n == 0 || (idx && keys)));
  return sdm_sort_by_key_async(ctx, idx, keys, n);
}

// ---------------------------------------------------------------------------------------
// remove_zero_n_or_flagged (collisions_methods.py:664-680).  The serial loop fills each dead
// slot of the surviving prefix [0, new_len) with the last live element still in the tail, taken
// from the end backwards: the r-th dead prefix slot (ascending) receives the r-th live tail
// element (descending); every tail slot ends up holding the sentinel.  Done here with prefix
// counts; wavefront ballot + popcount give the in-wave ranks.
// ctl words: [0] length in, [1] new length out, [2] number of prefix holes, [3] total dead
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool sd_dead(const int64_t *__restrict__ multiplicity,
                                        const int64_t *__restrict__ idx, int64_t i,
                                        int64_t flag) {
  const int64_t v = idx[i];
  return v == flag || multiplicity[v] == 0;
}

// Fused-step hooks: `fctl` (may be NULL) is the fused control block {valid, work, sorted,
// healthy, ...}; with it the compaction runs only if fctl[3] == 0 (unhealthy), over
// [0, fctl[0]) (particle_attributes.py:69), and its last block finally commits
// valid = work = new length, sorted = 0, healthy = 1 (single-cell: cell_start = {0, new length},
// sorted stays 1, since the sort of one cell is the identity).
#define FCTL_VALID 0
#define FCTL_WORK 1
#define FCTL_SORTED 2
#define FCTL_HEALTHY 3

// ---- one launch: exits at once while healthy; otherwise three phases (dead count per wavefront
// chunk; holes / fillers; apply) separated by a software grid barrier.  A wavefront owns a
// contiguous chunk of positions and walks it in tiles of SDM_WAVE (ballots only); after the first
// barrier every workgroup scans the COMPACT_WAVES chunk totals for itself (8 KB from L2 - cheaper
// than a fourth barrier around a scan by one workgroup).  The grid is ctx->compact_grid workgroups
// of COMPACT_THREADS: COMPACT_GRID, or as many as the occupancy query says are co-resident on this
// device (a partitioned or CU-masked device has fewer CUs), fixed at the context's first use.  A grid barrier is one same-address atomic per workgroup
// plus polling: measured 13.5 us with 256 workgroups, 5.1 us with 128, 3.0 us with 64 - hence
// few, large ones (128 x 1024 threads measured best end to end).
// Every spin is bounded (bar[2] is set on time-out).
#ifndef COMPACT_GRID
#define COMPACT_GRID 128
#endif
#ifndef COMPACT_THREADS
#define COMPACT_THREADS 1024
#endif
// capacity of the per-wavefront tables: the stand-alone kernel runs COMPACT_GRID workgroups, the
// prologue of k_bin_sort as many as that kernel has (at most COMPACT_MAX_GROUPS)
#define COMPACT_MAX_GROUPS 256
#define COMPACT_WAVES (COMPACT_MAX_GROUPS * COMPACT_THREADS / SDM_WAVE)
static_assert(COMPACT_GRID <= COMPACT_MAX_GROUPS && COMPACT_THREADS == BIN_THREADS, "shapes");

__device__ __forceinline__ bool grid_barrier(unsigned int *bar, unsigned int target) {
  __shared__ bool ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();  // release this workgroup's writes
    atomicAdd(&bar[0], 1u);
    unsigned int spins = 0;
    ok = true;
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) { ok = false; bar[2] = 1; break; }
    }
    __threadfence();  // acquire: drop stale L1 lines before reading the others' results
  }
  __syncthreads();
  return ok;
}

// ---- compaction from a LIST of the positions to remove ---------------------------------------
// Where the positions of the dead are known as a list (sharded steps: they were exchanged; adaptive
// steps of one cell: the kernels that flag them also list them) the reference's swap-from-the-end
// (collisions_methods.py:664-680; compact_run below) needs no pass over the permutation - the r-th
// hole of the surviving prefix (ascending) takes the r-th live element of the tail (from the end
// backwards), and only the d tail positions and the d listed ones are touched.  One workgroup of
// 1024; p: LDS for the next power of two >= d positions.  Leaves what compact_run leaves: holes[]
// (ascending) and ctl[2] = their number for the closed-form re-sort, the control words committed.
// 41-47 us -> ~10 us per removal at 2^22 positions.
__device__ __forceinline__ void
compact_listed_body(int32_t *p, int64_t *__restrict__ idx, const int64_t *__restrict__ dead,
                    int64_t d, int64_t flag, int64_t *__restrict__ fctl, int64_t *__restrict__ ctl,
                    int32_t *__restrict__ holes, int64_t *__restrict__ cell_start_single) {
  const int64_t length = fctl[FCTL_VALID], new_len = length - d;
  int n2 = 1;
  while (n2 < d) n2 <<= 1;
  for (int t = threadIdx.x; t < n2; t += 1024) p[t] = t < d ? (int32_t)dead[t] : INT32_MAX;
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)  // bitonic sort, ascending
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < n2; t += 1024) {
        const int o = t ^ j;
        if (o > t) {
          const int32_t a = p[t], b = p[o];
          if (((t & k) == 0) == (a > b)) { p[t] = b; p[o] = a; }
        }
      }
      __syncthreads();
    }
  // number of listed positions below x (the list is sorted)
  auto below = [&](int64_t x) -> int {
    int lo = 0, hi = (int)d;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int h = below(new_len);  // holes: the listed positions of the surviving prefix
  for (int t = threadIdx.x; t < d; t += 1024) {
    const int64_t x = length - 1 - t;  // the tail, from the end backwards
    const int b = below(x);
    const bool is_dead = b < d && p[b] == x;
    if (!is_dead) {
      const int r = t - ((int)d - below(x + 1));  // live tail elements above x
      const int64_t filler = idx[x];
      idx[p[r]] = filler;
    }
    if (t < h) holes[t] = p[t];
  }
  __syncthreads();  // (every filler was read before its tail slot takes the flag)
  for (int t = threadIdx.x; t < d; t += 1024) idx[new_len + t] = flag;
  if (threadIdx.x == 0) {
    ctl[0] = length;
    ctl[1] = new_len;
    ctl[2] = h;
    ctl[3] = d;
    fctl[FCTL_VALID] = new_len;
    fctl[FCTL_WORK] = new_len;
    fctl[FCTL_HEALTHY] = 1;
    if (cell_start_single) {
      cell_start_single[0] = 0;
      cell_start_single[1] = new_len;
    } else {
      fctl[FCTL_SORTED] = 0;
    }
  }
}

#define COMPACT_UNROLL 8
// "dead" of lane's position in COMPACT_UNROLL consecutive tiles (false beyond tile_end / length);
// val (optional): the idx values read
template <bool FLAG_ONLY>
__device__ __forceinline__ void compact_tiles(const int64_t *__restrict__ multiplicity,
                                              const int64_t *__restrict__ idx, int64_t flag,
                                              int64_t length, int tile, int tile_end, int lane,
                                              bool *dead, int64_t *val) {
  int64_t v[COMPACT_UNROLL], n[COMPACT_UNROLL];
  bool in[COMPACT_UNROLL];
#pragma unroll
  for (int u = 0; u < COMPACT_UNROLL; ++u) {
    const int64_t i = (int64_t)(tile + u) * SDM_WAVE + lane;
    in[u] = tile + u < tile_end && i < length;
    v[u] = in[u] ? idx[i] : flag;
  }
#pragma unroll
  for (int u = 0; u < COMPACT_UNROLL; ++u)
    n[u] = (FLAG_ONLY || v[u] == flag) ? 1 : multiplicity[v[u]];
#pragma unroll
  for (int u = 0; u < COMPACT_UNROLL; ++u) {
    dead[u] = in[u] && (v[u] == flag || n[u] == 0);
    if (val) val[u] = v[u];
  }
}

// called by every thread of one workgroup (>= SDM_CNT_SLOTS threads) once the control words are final
__device__ __forceinline__ void compact_epilogue(const CompactEpilogue &E,
                                                 int64_t *__restrict__ fctl) {
  if (E.slots) {
    __shared__ int64_t part[SDM_CNT_SLOTS / SDM_WAVE];
    if (threadIdx.x < SDM_CNT_SLOTS) {
      int64_t *word = E.slots + threadIdx.x * SDM_CNT_STRIDE + SDM_CNT_OVERFLOW;
      const int64_t v = *word;
      if (v != 0) *word = 0;
      const int64_t s = wave_sum_i64(v);
      if (lane_id() == 0) part[threadIdx.x / SDM_WAVE] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int64_t all = 0;
      for (int w = 0; w < SDM_CNT_SLOTS / SDM_WAVE; ++w) all += part[w];
      if (all != 0) fctl[4] += all;
    }
  }
  if (threadIdx.x == 0) {
    // device side: the working length of one cell is the valid length whether the time step goes
    // on or ends (reset_working_length, collision.py:189); only the host is told "0 = done"
    fctl[FCTL_WORK] = fctl[FCTL_VALID];
    const double left = E.dt_left[0];
    if (E.dt_left_pub) E.dt_left_pub[0] = left;
    publish_ctl(fctl, E.box, E.seq, left != 0 ? fctl[FCTL_VALID] : 0);
  }
}

// FLAG_ONLY: the caller guarantees that no live super-droplet has zero multiplicity (it entered
// with a healthy state and only flags positions), so the random gather of multiplicities is skipped
// the compaction proper, by every workgroup of the grid (all resident; at most COMPACT_MAX_GROUPS);
// false: a grid barrier timed out (fctl[7] = 2).  *new_length: the length every workgroup computed
template <bool FLAG_ONLY>
__device__ __forceinline__ bool
compact_run(const int64_t *__restrict__ multiplicity, int64_t *__restrict__ idx, int64_t flag,
            int64_t *__restrict__ fctl, int32_t *__restrict__ wave_dead, int n_tiles,
            int64_t *__restrict__ ctl, int32_t *__restrict__ holes, int64_t *__restrict__ fillers,
            int64_t *__restrict__ cell_start_single, unsigned int *__restrict__ bar,
            const CompactEpilogue &E, int64_t *new_length, int *excl) {
  const int64_t length = fctl[FCTL_VALID];
  // (bar[3]: the barrier k_bin_build2 passes after its re-sort; nobody touches it before this
  // run's last barrier, every workgroup passes the first one after this store)
  if (blockIdx.x == 0 && threadIdx.x == 0) bar[3] = 0;
  // excl: COMPACT_WAVES ints of LDS from the caller (16 KB: k_bin_build2 lends its dynamic memory,
  // so that the code of this rare path does not cost it a workgroup per CU)
  __shared__ int sm[COMPACT_THREADS / SDM_WAVE];
  const int lane = lane_id(), w = threadIdx.x / SDM_WAVE;
  const int wave = blockIdx.x * (COMPACT_THREADS / SDM_WAVE) + w;
  const int n_waves = (int)gridDim.x * (COMPACT_THREADS / SDM_WAVE);  // <= COMPACT_WAVES
  const int per_wave = (n_tiles + n_waves - 1) / n_waves;
  const int tile0 = wave * per_wave, tile1 = min(n_tiles, tile0 + per_wave);
  // phase A: dead count of every wavefront's chunk
  // (COMPACT_UNROLL tiles per round: their loads are in flight together - a wavefront walking its
  // chunk one dependent load at a time is latency-bound)
  {
    int count = 0;
    for (int tile = tile0; tile < tile1; tile += COMPACT_UNROLL) {
      bool dead[COMPACT_UNROLL];
      compact_tiles<FLAG_ONLY>(multiplicity, idx, flag, length, tile, tile1, lane, dead, nullptr);
#pragma unroll
      for (int u = 0; u < COMPACT_UNROLL; ++u) count += __popcll(__ballot(dead[u]));
    }
    if (lane == 0) wave_dead[wave] = count;
  }
#define BARRIER_OR_FAIL(k)                                   \
  if (!grid_barrier(bar, (k) * gridDim.x)) {                 \
    if (threadIdx.x == 0) fctl[7] = 2; /* surfaced by the host as an error */ \
    return false;                                            \
  }
  BARRIER_OR_FAIL(1)
  // exclusive scan of the chunk totals, by every workgroup for itself
  int64_t total_dead;
  {
    constexpr int PER = (COMPACT_WAVES + COMPACT_THREADS - 1) / COMPACT_THREADS;
    const int b0 = threadIdx.x * PER;
    int v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      v[k] = b0 + k < n_waves ? ((volatile int32_t *)wave_dead)[b0 + k] : 0;
      sum += v[k];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    int base = 0, all = 0;
    for (int k = 0; k < COMPACT_THREADS / SDM_WAVE; ++k) {
      if (k < w) base += sm[k];
      all += sm[k];
    }
    int run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (b0 + k < n_waves) {
        excl[b0 + k] = run;
        run += v[k];
      }
    __syncthreads();
    total_dead = all;
  }
  const int64_t new_len = length - total_dead;
  *new_length = new_len;
  // phase C: holes of the surviving prefix, live elements of the tail (from the end backwards)
  if (total_dead != 0) {
    int64_t before = excl[wave];  // dead positions ahead of the current tile
    for (int tile = tile0; tile < tile1; tile += COMPACT_UNROLL) {
      bool dead[COMPACT_UNROLL];
      int64_t val[COMPACT_UNROLL];
      compact_tiles<FLAG_ONLY>(multiplicity, idx, flag, length, tile, tile1, lane, dead, val);
#pragma unroll
      for (int u = 0; u < COMPACT_UNROLL; ++u) {
        const int64_t i = (int64_t)(tile + u) * SDM_WAVE + lane;
        const bool in = tile + u < tile1 && i < length;
        const unsigned long long m = __ballot(dead[u]);
        const int64_t dp = before + __popcll(m & ((1ull << lane) - 1));
        before += __popcll(m);
        if (in) {
          if (i == new_len) ctl[2] = dp;
          if (i < new_len) {
            if (dead[u]) holes[dp] = (int32_t)i;
          } else if (!dead[u]) {
            fillers[(length - 1 - i) - (total_dead - dp)] = val[u];
          }
        }
      }
    }
  }
  BARRIER_OR_FAIL(2)
  // phase D: apply
  if (total_dead != 0) {
    const int64_t n_holes = ((volatile int64_t *)ctl)[2];
    for (int64_t t = (int64_t)blockIdx.x * COMPACT_THREADS + threadIdx.x; t < length - new_len;
         t += (int64_t)gridDim.x * COMPACT_THREADS) {
      idx[new_len + t] = flag;
      if (t < n_holes) idx[((volatile int32_t *)holes)[t]] = ((volatile int64_t *)fillers)[t];
    }
  }
  BARRIER_OR_FAIL(3)
#undef BARRIER_OR_FAIL
  // every workgroup is past the last barrier once it arrives here: the last one re-arms the
  // barrier words and commits the control words
  __shared__ bool last;
  if (threadIdx.x == 0) last = atomicAdd(&bar[1], 1u) == gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x == 0) {
    bar[0] = 0;
    bar[1] = 0;
    fctl[FCTL_VALID] = new_len;
    fctl[FCTL_WORK] = new_len;
    fctl[FCTL_HEALTHY] = 1;
    if (cell_start_single) {
      cell_start_single[0] = 0;
      cell_start_single[1] = new_len;
    } else {
      fctl[FCTL_SORTED] = 0;
    }
  }
  if (last && E.dt_left) compact_epilogue(E, fctl);
  return true;
}

template <bool FLAG_ONLY>
__global__ void __launch_bounds__(COMPACT_THREADS)
k_compact_persistent(const int64_t *__restrict__ multiplicity, int64_t *__restrict__ idx,
                     int64_t flag, int64_t *__restrict__ fctl, int32_t *__restrict__ wave_dead,
                     int n_tiles, int64_t *__restrict__ ctl, int32_t *__restrict__ holes,
                     int64_t *__restrict__ fillers, int64_t *__restrict__ cell_start_single,
                     unsigned int *__restrict__ bar, CompactEpilogue E) {
  if (E.gwords && blockIdx.x == 0 && threadIdx.x == 0) {  // this sub-step's draws are consumed
    E.gwords[0] += E.advance;
    E.gwords[1] += E.advance_b;
  }
  // (the other counter: nobody reads it in this launch, the next sub-step's kernels add to it)
  if (E.dead_count_next && blockIdx.x == 0 && threadIdx.x == 0) *E.dead_count_next = 0;
  if (fctl[FCTL_HEALTHY] != 0) {
    if (E.dt_left && blockIdx.x == 0) compact_epilogue(E, fctl);
    return;
  }
  int64_t new_len;
  __shared__ int excl[COMPACT_WAVES];
  if (E.dead_count) {
    const unsigned long long d = *E.dead_count;  // the same for every workgroup: written before
    if (d >= 1 && d <= COMPACT_WAVES) {          // this launch, cleared by the next one
      if (blockIdx.x != 0) return;
      compact_listed_body((int32_t *)excl, idx, E.dead_pos, (int64_t)d, flag, fctl, ctl, holes,
                          cell_start_single);
      __syncthreads();
      if (E.dt_left) compact_epilogue(E, fctl);
      return;
    }
  }
  (void)compact_run<FLAG_ONLY>(multiplicity, idx, flag, fctl, wave_dead, n_tiles, ctl, holes,
                               fillers, cell_start_single, bar, E, &new_len, excl);
}

// k_bin_sort (declared above): the tile sort of the shuffle build
template <bool RNG, int TILE>
__global__ void __launch_bounds__(BIN_THREADS)
k_bin_sort(int2 *__restrict__ events, int32_t *__restrict__ toff, int32_t *__restrict__ jarr,
           int32_t *__restrict__ loc, int n_bins, const double *__restrict__ u01,
           const int64_t *__restrict__ cell_start,
           int64_t n_cell, const int64_t *__restrict__ p_length, int64_t length_arg, u128 s_off,
           u128 inc, const u128 *__restrict__ tab, const uint64_t *__restrict__ dev_off,
           const u128 *__restrict__ aff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t length = p_length ? *p_length : length_arg;
  bin_sort_body<RNG, TILE>(smem, events, toff, jarr, loc, n_bins, u01, cell_start, n_cell, length, -1,
                     s_off, inc, tab, dev_off, aff);
}

template <int FMT>
__global__ void __launch_bounds__(BIN_THREADS)
k_bin_build2(void *__restrict__ rec_out, int32_t *__restrict__ ovf_head,
             int32_t *__restrict__ ovf_next, const int2 *__restrict__ events,
             const int32_t *__restrict__ toff, const int32_t *__restrict__ jarr, int n_bins,
             int n_tiles, int ev_tile, const int64_t *__restrict__ idx0,
             const int64_t *__restrict__ p_length, int64_t length_arg, BuildPrologue P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (SDM_REC_CHAIN: five - a position with more hits than inline slots sends every thread that
  // asks about it through a list in global memory, and with four slots nearly every wavefront
  // holds such a position: 0.37 % of the positions, 0.06 % with five)
  constexpr int SLOTS = FMT == SDM_REC_CHAIN ? 5 : FMT == SDM_REC_P21 ? 4 : (FMT == SDM_REC_P24 ? 3 : 2);
  int32_t *slot = (int32_t *)smem;  // SLOTS x BIN_POS, then BIN_POS list heads
  int32_t *head = slot + SLOTS * BIN_POS;
  // SDM_REC_CHAIN: own target / id / place in the sorted array of every position of the bin
  int32_t *jp_l = head + BIN_POS, *id_l = jp_l + BIN_POS, *loc_l = id_l + BIN_POS;
  // P.compact.fctl: the events were sorted ahead, by the pair kernel of the previous sub-step, for
  // the length that sub-step began with.  If a super-droplet died in it (rare), the compaction
  // runs here, every workgroup sorts its tile again for the new length (n_tiles == n_bins on this
  // route) and a further grid barrier separates that from the gathering of the runs
  int64_t length;
  if (P.compact.fctl && P.compact.fctl[FCTL_HEALTHY] == 0) {
    const CompactEpilogue none = {nullptr, nullptr, nullptr, 0, nullptr, 0, 0};
    const SortPrologue &C = P.compact;
    if (!compact_run<true>(C.multiplicity, C.idx, C.flag, C.fctl, C.wave_dead, C.n_tiles, C.ctl,
                           C.holes, C.fillers, C.cell_start_single, C.bar, none, &length,
                           (int *)smem))
      return;
    __syncthreads();
    bin_sort_body<true>(smem, P.events, P.toff, P.jarr, P.loc, n_bins, nullptr, nullptr, 1, length,
                        length, P.s_off, P.inc, P.tab, nullptr, P.aff);
    if (!grid_barrier(C.bar + 3, gridDim.x)) {
      if (threadIdx.x == 0) C.fctl[7] = 2;
      return;
    }
  } else {
    length = p_length ? *p_length : length_arg;
  }
  const int64_t base = (int64_t)blockIdx.x * BIN_POS;
  if (base >= length) return;
  BIN_MARK(8);
  const int bin = blockIdx.x;
  constexpr int PER_POS = BIN_POS / BIN_THREADS;
  // this bin's run in every tile's segment: `tpt` threads share a tile (4 at 2^20
  // super-droplets: runs hold ~16 events), each takes every tpt-th event of the run; the first
  // RUN_AHEAD of them are requested before any is placed.
  // Order of the requests: the kernel is two dependent memory round trips (a run's bounds, then its
  // events) around LDS work, and loads return in the order they were issued - so the bounds go
  // first, the slot initialisation runs while they travel, the events of the thread's first tile
  // are requested BEFORE the barrier, and what is only needed at the end comes last
  const int tpt = n_tiles >= BIN_THREADS ? 1 : BIN_THREADS / n_tiles;
  const int sub = threadIdx.x % tpt, t_step = BIN_THREADS / tpt;
  const int t_first = threadIdx.x / tpt;
  int a_first = 0, b_first = 0;
  if (t_first < n_tiles) {
    const int32_t *row = toff + (int64_t)t_first * (n_bins + 1) + bin;
    a_first = row[0];
    b_first = row[1];
  }
  // (16 bytes per store: the slots and list heads are 96 KB with five slots)
  for (int q = threadIdx.x; q < (SLOTS + 1) * BIN_POS / 4; q += BIN_THREADS)
    ((int4 *)slot)[q] = make_int4(-1, -1, -1, -1);
  constexpr int RUN_AHEAD = 8;
  // (the first RUN_AHEAD events of the thread's FIRST tile - at 2^20 positions its only one - stay
  // in registers: the S pass below needs the same events again)
  int2 ev0[RUN_AHEAD];
  {
    const int2 *run = events + (int64_t)t_first * ev_tile;
#pragma unroll
    for (int k = 0; k < RUN_AHEAD; ++k) {
      const int x = a_first + sub + k * tpt;
      ev0[k] = x < b_first ? run[x] : make_int2(-1, 0);  // (b_first = 0 beyond the last tile)
    }
  }
  // what the records need from memory besides the hits, requested now, used at the end
  int64_t id_v[PER_POS];
  int32_t j_v[PER_POS], loc_v[PER_POS];
#pragma unroll
  for (int k = 0; k < PER_POS; ++k) {
    const int64_t p = base + threadIdx.x + k * BIN_THREADS;
    id_v[k] = p < length ? idx0[p] : 0;
    j_v[k] = p < length ? jarr[p] : -1;
    loc_v[k] = (FMT == SDM_REC_CHAIN && p < length) ? P.loc[p] : -1;
  }
  BIN_MARK(9);
  // LDS only: __syncthreads() would also wait for the global loads requested above
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  BIN_MARK(10);
  // hits on this bin's positions: the first SLOTS inline (claimed by compare-and-swap), rest listed
  auto place = [&](int2 e) {
    const int q = e.y - (int)base;
    bool placed = false;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k)
      if (!placed) placed = atomicCAS(&slot[k * BIN_POS + q], -1, e.x) == -1;
    // (-1 terminated, built entirely here; SDM_REC_CHAIN keeps the links in scratch of their own:
    // ovf_next is its ssucc table)
    if (!placed)
      (FMT == SDM_REC_CHAIN ? P.chain_links : ovf_next)[e.x] = atomicExch(&head[q], e.x);
  };
  for (int t = t_first; t < n_tiles; t += t_step) {
    int a = a_first, b = b_first;
    if (t != t_first) {
      const int32_t *row = toff + (int64_t)t * (n_bins + 1) + bin;
      a = row[0];
      b = row[1];
    }
    const int2 *run = events + (int64_t)t * ev_tile;  // (tile-major: the tile's own segment)
    int2 ev[RUN_AHEAD];
#pragma unroll
    for (int k = 0; k < RUN_AHEAD; ++k) {
      const int x = a + sub + k * tpt;
      ev[k] = t == t_first ? ev0[k] : (x < b ? run[x] : make_int2(-1, 0));
    }
#pragma unroll
    for (int k = 0; k < RUN_AHEAD; ++k)
      if (ev[k].x >= 0) place(ev[k]);
    for (int x = a + sub + RUN_AHEAD * tpt; x < b; x += tpt) place(run[x]);
  }
  if (FMT == SDM_REC_CHAIN) {
    // (the loads requested at the kernel's start have long arrived)
#pragma unroll
    for (int k = 0; k < PER_POS; ++k) {
      const int q = threadIdx.x + k * BIN_THREADS;
      jp_l[q] = j_v[k];
      id_l[q] = (int32_t)id_v[k];
      loc_l[q] = loc_v[k];
    }
  }
  __syncthreads();
  BIN_MARK(11);
  // SDM_REC_CHAIN (shuffle_device.h): everything that touches a position of this bin is in LDS now -
  // its own event (jp_l >= 0: it has one) and the events that hit it (inline slots, the overflow
  // list this workgroup has just built).  F(q, e): what the content of position q is once all
  // events above e have been applied, as a successor word
  int32_t *links = P.chain_links;
  auto hit_above = [&](int q, int32_t e) -> int32_t {  // smallest hit on q with index > e
    int32_t best = INT32_MAX;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int32_t v = slot[k * BIN_POS + q];
      if (v > e && v < best) best = v;
    }
    for (int32_t t = head[q]; t >= 0; t = links[t])
      if (t > e && t < best) best = t;
    return best;
  };
  auto value_after = [&](int q, int32_t e) -> uint32_t {
    const int32_t up = hit_above(q, e);
    const int32_t pi = (int32_t)base + q;
    const int32_t own = (jp_l[q] >= 0 && pi > e) ? pi : INT32_MAX;
    if (up == INT32_MAX && own == INT32_MAX) return chain_word(0, id_l[q]);
    return own <= up ? chain_word(CHAIN_S, loc_l[q]) : chain_word(CHAIN_T, up);
  };
  if (FMT == SDM_REC_CHAIN) {
    // the S word of every event that hits this bin, written where the event was read: the same
    // runs of the tile-sorted array, the same threads
    for (int t = t_first; t < n_tiles; t += t_step) {
      int a = a_first, b = b_first;
      if (t != t_first) {
        const int32_t *row = toff + (int64_t)t * (n_bins + 1) + bin;
        a = row[0];
        b = row[1];
      }
      const int2 *run = events + (int64_t)t * ev_tile;
      uint32_t *out = P.ssucc + (int64_t)t * ev_tile;
      int x = a + sub;
      if (t == t_first) {  // from the registers of the placing pass
#pragma unroll
        for (int k = 0; k < RUN_AHEAD; ++k) {
          if (ev0[k].x >= 0) out[x] = value_after(ev0[k].y - (int)base, ev0[k].x);
          x += tpt;
        }
      }
      for (; x < b; x += tpt) {
        const int2 e = run[x];
        out[x] = value_after(e.y - (int)base, e.x);
      }
    }
  }
  BIN_MARK(13);
#pragma unroll
  for (int kq = 0; kq < PER_POS; ++kq) {
    const int q = threadIdx.x + kq * BIN_THREADS;
    const int64_t p = base + q;
    if (p >= length) break;
    const int32_t h = head[q];
    const int32_t id = (int32_t)id_v[kq];
    const int32_t jp = j_v[kq];
    if (FMT == SDM_REC_CHAIN) {
      uint32_t *first = (uint32_t *)rec_out, *tsucc = (uint32_t *)ovf_head;
      first[p] = value_after(q, -1);
      const int32_t up = hit_above(q, (int32_t)p);
      tsucc[p] = up == INT32_MAX ? chain_word(0, id) : chain_word(CHAIN_T, up);
    } else if (FMT == SDM_REC_P21) {
      PackRec21 r;
      p21_pack(r.lo, r.hi, jp, slot[q], slot[BIN_POS + q], slot[2 * BIN_POS + q],
               slot[(SLOTS - 1) * BIN_POS + q], id, h >= 0);
      ((PackRec21 *)rec_out)[p] = r;
    } else if (FMT == SDM_REC_P24) {
      PackRec24 r;
      p24_pack(r.lo, r.hi, jp, slot[q], slot[BIN_POS + q], slot[2 * BIN_POS + q], id, h >= 0);
      ((PackRec24 *)rec_out)[p] = r;
    } else {
      PackRec r;
      r.j = jp;
      r.s0 = slot[q];
      r.s1 = slot[BIN_POS + q];
      r.val = id | (h >= 0 ? (int32_t)0x80000000 : 0);
      ((PackRec *)rec_out)[p] = r;
    }
    if (FMT != SDM_REC_CHAIN && h >= 0) ovf_head[p] = h;
  }
  BIN_MARK(12);
}

// `bar`: 4 zero-initialised device words owned by the caller (persist across launches)
int sdm_compact_fused_async(sdm_ctx *ctx, char *scratch, const int64_t *multiplicity,
                            int64_t *idx, int64_t length_bound, int64_t flag, int64_t *fctl,
                            int64_t *ctl, int64_t *cell_start_single, bool flag_only,
                            const CompactEpilogue *epilogue) {
  CompactEpilogue E;
  memset(&E, 0, sizeof(E));
  if (epilogue) E = *epilogue;
  Carver cv(scratch);
  const int n_tiles = (int)grid_for(length_bound, SDM_WAVE);
  int32_t *wave_dead = cv.take<int32_t>(COMPACT_WAVES);
  int32_t *holes = cv.take<int32_t>(length_bound);
  int64_t *fillers = cv.take<int64_t>(length_bound);
  unsigned int *bar = (unsigned int *)(ctx->dscal + 12);
  if (ctx->compact_grid == 0) {
    // the software grid barrier needs every workgroup resident at once
    int per_cu = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_compact_persistent<false>,
                                                         COMPACT_THREADS, 0));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const int fit = per_cu * cus;
    if (fit < 1) {
      sdm_set_error("the compaction kernel does not fit this device (occupancy query: %d x %d)",
                    per_cu, cus);
      return SDM_E_HIP;
    }
    ctx->compact_grid = fit < COMPACT_GRID ? fit : COMPACT_GRID;
  }
  const unsigned grid = (unsigned)ctx->compact_grid;
  if (flag_only)
    hipLaunchKernelGGL(k_compact_persistent<true>, dim3(grid), dim3(COMPACT_THREADS), 0,
                       ctx->stream, multiplicity, idx, flag, fctl, wave_dead, n_tiles, ctl, holes,
                       fillers, cell_start_single, bar, E);
  else
    hipLaunchKernelGGL(k_compact_persistent<false>, dim3(grid), dim3(COMPACT_THREADS), 0,
                       ctx->stream, multiplicity, idx, flag, fctl, wave_dead, n_tiles, ctl, holes,
                       fillers, cell_start_single, bar, E);
  LAUNCH_CHECK();
  return SDM_OK;
}

// the list handed over by the host: d <= COMPACT_LIST_CAP (more: the full kernel)
#define COMPACT_LIST_CAP 8192
__global__ void __launch_bounds__(1024)
k_compact_listed(int64_t *__restrict__ idx, const int64_t *__restrict__ dead, int64_t d,
                 int64_t flag, int64_t *__restrict__ fctl, int64_t *__restrict__ ctl,
                 int32_t *__restrict__ holes, int64_t *__restrict__ cell_start_single) {
  __shared__ int32_t p[COMPACT_LIST_CAP];
  compact_listed_body(p, idx, dead, d, flag, fctl, ctl, holes, cell_start_single);
}

// `dead`: d distinct positions < fctl[0], all flagged in idx already (device list); scratch and
// ctl as for sdm_compact_fused_async (the closed-form re-sort reads holes[] and ctl[2] from there)
int sdm_compact_listed_async(sdm_ctx *ctx, char *scratch, int64_t *idx, const int64_t *dead,
                             int64_t d, int64_t length_bound, int64_t flag, int64_t *fctl,
                             int64_t *ctl, int64_t *cell_start_single) {
  ARG_TRY(d >= 1 && d <= COMPACT_LIST_CAP);
  Carver cv(scratch);
  (void)cv.take<int32_t>(COMPACT_WAVES);
  int32_t *holes = cv.take<int32_t>(length_bound);
  hipLaunchKernelGGL(k_compact_listed, dim3(1), dim3(1024), 0, ctx->stream, idx, dead, d, flag,
                     fctl, ctl, holes, cell_start_single);
  LAUNCH_CHECK();
  return SDM_OK;
}
bool sdm_compact_listed_fits(int64_t d) { return d >= 1 && d <= COMPACT_LIST_CAP; }

bool sdm_shuffle_presort_ok(sdm_ctx *ctx, int64_t length_bound, int64_t id_bound) {
  if (!binned_ok(length_bound, false)) return false;
  const int nb = bin_count(length_bound), nt = ev_tile_count(length_bound);
  const int64_t both = id_bound > length_bound ? id_bound : length_bound;
  if (nb != nt || nb > COMPACT_MAX_GROUPS || id_bound < 0 || both > P21_MAX) return false;
  if (ctx->build_resident == 0) {
    const bool chain = chain_enabled(ctx) && nb * 16 <= EV_TILE;
    const size_t lds_build = sizeof(int32_t) * (size_t)((chain ? 9 : 5) * BIN_POS);
    int per_cu = 0, cus = 0;
    const void *kernel = chain ? (const void *)k_bin_build2<SDM_REC_CHAIN>
                                         : (const void *)k_bin_build2<SDM_REC_P21>;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(lds_build > 65536 ? lds_build : 65536)) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BIN_THREADS, lds_build) !=
            hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) !=
            hipSuccess) {
      (void)hipGetLastError();
      ctx->build_resident = -1;
    } else {
      ctx->build_resident = per_cu * cus > 0 ? per_cu * cus : -1;
    }
  }
  return nb <= ctx->build_resident;
}

// the arguments of sdm_compact_fused_async(.., flag_only = true) for the prologue of the next
// build (sdm_shuffle_presort_ok)
void sdm_compact_as_prologue(sdm_ctx *ctx, char *scratch, const int64_t *multiplicity,
                             int64_t *idx, int64_t length_bound, int64_t flag, int64_t *fctl,
                             int64_t *ctl, int64_t *cell_start_single, SortPrologue *out) {
  Carver cv(scratch);
  out->fctl = fctl;
  out->multiplicity = multiplicity;
  out->idx = idx;
  out->flag = flag;
  out->n_tiles = (int)grid_for(length_bound, SDM_WAVE);
  out->wave_dead = cv.take<int32_t>(COMPACT_WAVES);
  out->holes = cv.take<int32_t>(length_bound);
  out->fillers = cv.take<int64_t>(length_bound);
  out->ctl = ctl;
  out->cell_start_single = cell_start_single;
  out->bar = (unsigned int *)(ctx->dscal + 12);
}

// after a timed-out grid barrier (error word 2) the workgroups returned without re-arming the
// barrier words: clear them, so that the context stays usable
int sdm_compact_rearm(sdm_ctx *ctx) {
  HIP_TRY(hipMemsetAsync(ctx->dscal + 12, 0, sizeof(int64_t) * 4, ctx->stream));
  return SDM_OK;
}

// ---- re-sorting by cell after a compaction, without sorting ----------------------------------
// A state that was sorted by cell when super-droplets died is, after the reference's
// swap-from-the-end (holes filled with the live tail, compact_run above), still GROUPED by cell
// except for the fillers; and if the whole tail lay in the last non-empty segment (a handful of
// deaths against thousands of members: nearly always) the fillers all belong to that segment's
// cell.  The stable counting sort that the cell_start getter would run then has a closed form:
//   * the members of a segment keep their order; a filler standing in an EARLIER segment's hole
//     precedes everything that stands in the last segment (it comes first in the permutation),
//     fillers among themselves in hole order; holes inside the last segment were refilled with
//     members of the same cell and change nothing;
//   * the segments themselves are re-ordered: the sort runs under the cell_idx of the CURRENT
//     sub-step (cell_idx.sort_by_key(dt_left) runs every sub-step, the permutation keeps the
//     order of the last sort), so segment s goes where the key of its cell says.
// Sizes per segment -> scan in key order -> one scatter pass -> copy back: streaming, no
// histogram of 2^22 gathered keys.  Everything is decided on the device (plan[0]); when the form
// does not apply, the counting sort that follows finds the state unsorted and runs as before.
#define RESORT_CAP 4096
__global__ void k_sort_cellstart(const int64_t *__restrict__ count, int64_t *__restrict__ cell_start,
                                 int64_t n_cell, const int64_t *__restrict__ p_length);
__global__ void k_resort_plan(const int64_t *__restrict__ fctl, const int64_t *__restrict__ cctl,
                              const int64_t *__restrict__ cell_start, int64_t n_cell,
                              const int32_t *__restrict__ holes, int64_t *__restrict__ plan) {
  // (one wavefront: the count of the holes before the last segment is a reduction, not a walk)
  const int64_t old_len = cell_start[n_cell], new_len = fctl[FCTL_VALID], n_holes = cctl[2];
  bool ok = fctl[FCTL_SORTED] == 0 && fctl[FCTL_WORK] == new_len && old_len > new_len &&
            new_len > 0 && n_holes >= 0 && n_holes <= RESORT_CAP && cell_start[0] == 0;
  int64_t s_last = 0, cs_last = 0;
  int h0 = 0;
  if (ok) {
    s_last = find_cell(cell_start, n_cell, old_len - 1);
    cs_last = cell_start[s_last];
    ok = cs_last <= new_len;  // the whole tail [new_len, old_len) lay in that segment
    if (ok)
      for (int k = threadIdx.x; k < (int)n_holes; k += SDM_WAVE) h0 += holes[k] < cs_last;
  }
  h0 = wave_sum_i32(h0);
  if (threadIdx.x != 0) return;
  plan[0] = ok ? 1 : 0;
  plan[1] = s_last;
  plan[2] = cs_last;
  plan[3] = n_holes;
  plan[4] = new_len;
  plan[5] = h0;
}
__device__ __forceinline__ int holes_before(const int32_t *h, int n, int64_t x) {  // # holes < x
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (h[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// one thread per segment: its size after the removal, and the key of its cell under the current
// cell_idx; seg_size (zeroed by the caller) is indexed by that key, seg_key by the segment
__global__ void __launch_bounds__(SDM_BLOCK)
k_resort_segments(const int64_t *__restrict__ plan, const int64_t *__restrict__ idx,
                  const int32_t *__restrict__ holes, const int64_t *__restrict__ cell_start,
                  const int64_t *__restrict__ cell_id, const int64_t *__restrict__ cell_idx,
                  int64_t n_cell, int64_t *__restrict__ seg_size, int64_t *__restrict__ seg_key) {
  if (plan[0] == 0) return;
  __shared__ int32_t sh[RESORT_CAP];
  const int64_t s_last = plan[1], cs_last = plan[2], new_len = plan[4], h0 = plan[5];
  const int n_holes = (int)plan[3];
  for (int k = threadIdx.x; k < n_holes; k += SDM_BLOCK) sh[k] = holes[k];
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (s >= n_cell) return;
  int64_t size = 0, member = -1;
  if (s < s_last) {
    const int64_t a = cell_start[s], b = cell_start[s + 1];
    int hb = holes_before(sh, n_holes, a);
    size = (b - a) - (holes_before(sh, n_holes, b) - hb);
    if (size > 0) {  // the first position of the segment that is not a hole (= not a filler)
      int64_t p = a;
      while (hb < n_holes && sh[hb] == p) { ++p; ++hb; }
      member = p;
    }
  } else if (s == s_last) {
    size = new_len - cs_last + h0;
    if (size > 0) member = new_len > cs_last ? cs_last : (int64_t)sh[0];  // (a filler: same cell)
  }
  int64_t key = -1;
  if (size > 0) {
    key = cell_idx[cell_id[idx[member]]];
    seg_size[key] = size;
  }
  seg_key[s] = key;
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_resort_scatter(const int64_t *__restrict__ plan, const int64_t *__restrict__ idx,
                 const int32_t *__restrict__ holes, int64_t *__restrict__ out,
                 const int64_t *__restrict__ cell_start, const int64_t *__restrict__ cs_new,
                 const int64_t *__restrict__ seg_key, int64_t n_cell) {
  if (plan[0] == 0) return;
  __shared__ int32_t sh[RESORT_CAP];
  const int64_t s_last = plan[1], cs_last = plan[2], new_len = plan[4], h0 = plan[5];
  const int n_holes = (int)plan[3];
  for (int k = threadIdx.x; k < n_holes; k += SDM_BLOCK) sh[k] = holes[k];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  // the segment of the wavefront's first position serves all its lanes unless a boundary falls
  // into the 64 positions (one look-up per wavefront instead of ten dependent loads per lane)
  const int64_t w0 = i - lane_id();
  int64_t s = 0, a = 0, b = 0;
  if (w0 < cs_last) {
    s = find_cell(cell_start, n_cell, w0);
    a = cell_start[s];
    b = cell_start[s + 1];
  }
  if (i >= new_len) return;
  const int hb = holes_before(sh, n_holes, i);
  const bool filler = hb < n_holes && sh[hb] == i;
  int64_t dst;
  if (i >= cs_last) {
    dst = cs_new[seg_key[s_last]] + h0 + (i - cs_last);
  } else if (filler) {
    dst = cs_new[seg_key[s_last]] + hb;
  } else {
    if (i >= b) {  // (a boundary inside the wavefront's range)
      s = find_cell(cell_start, n_cell, i);
      a = cell_start[s];
    }
    dst = cs_new[seg_key[s]] + (i - a) - (hb - holes_before(sh, n_holes, a));
  }
  out[dst] = idx[i];
}
__global__ void __launch_bounds__(SDM_BLOCK)
k_resort_commit(const int64_t *__restrict__ plan, int64_t *__restrict__ idx,
                const int64_t *__restrict__ out, int64_t *__restrict__ cell_start,
                const int64_t *__restrict__ cs_new, int64_t n_cell, int64_t *__restrict__ fctl) {
  if (plan[0] == 0) return;
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i < plan[4]) idx[i] = out[i];
  if (i <= n_cell) cell_start[i] = cs_new[i];
  if (i == 0) fctl[FCTL_SORTED] = 1;
}

// `scratch`, `cctl`: what the compaction just used (sdm_compact_fused_async); `plan`: 8 device
// words.  *applies: whether the closed form holds (decided on the device, read back: the host is
// about to wait for this stream anyway, and a counting sort that is launched only to find its
// gate closed still costs 60 us of empty grids at 2^22)
int sdm_resort_plan(sdm_ctx *ctx, char *scratch, int64_t length_bound, const int64_t *cctl,
                    const int64_t *fctl, const int64_t *cell_start, int64_t n_cell, int64_t *plan,
                    bool *applies) {
  Carver cv(scratch);
  (void)cv.take<int32_t>(COMPACT_WAVES);
  const int32_t *holes = cv.take<int32_t>(length_bound);
  hipLaunchKernelGGL(k_resort_plan, dim3(1), dim3(SDM_WAVE), 0, ctx->stream, fctl, cctl,
                     cell_start, n_cell, holes, plan);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, plan, sizeof(int64_t), hipMemcpyDeviceToHost,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *applies = ctx->mailbox[0] != 0;
  return SDM_OK;
}

// after sdm_resort_plan said it applies.  `seg_size`, `seg_key`: n_cell words each; `out`:
// length_bound words; `cs_new`: n_cell + 1
int sdm_resort_after_compaction_async(sdm_ctx *ctx, char *scratch, int64_t length_bound,
                                      int64_t *fctl, int64_t *idx, int64_t *out,
                                      int64_t *cell_start, int64_t *cs_new,
                                      const int64_t *cell_id, const int64_t *cell_idx,
                                      int64_t n_cell, const int64_t *plan, int64_t *seg_size,
                                      int64_t *seg_key) {
  Carver cv(scratch);
  (void)cv.take<int32_t>(COMPACT_WAVES);
  const int32_t *holes = cv.take<int32_t>(length_bound);
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemsetAsync(seg_size, 0, sizeof(int64_t) * (size_t)n_cell, s));
  hipLaunchKernelGGL(k_resort_segments, dim3(grid_for(n_cell)), dim3(SDM_BLOCK), 0, s, plan,
                     (const int64_t *)idx, holes, (const int64_t *)cell_start, cell_id, cell_idx,
                     n_cell, seg_size, seg_key);
  hipLaunchKernelGGL(k_sort_cellstart, dim3(1), dim3(1024), 0, s, (const int64_t *)seg_size,
                     cs_new, n_cell, plan);
  hipLaunchKernelGGL(k_resort_scatter, dim3(grid_for(length_bound)), dim3(SDM_BLOCK), 0, s, plan,
                     (const int64_t *)idx, holes, out, (const int64_t *)cell_start,
                     (const int64_t *)cs_new, (const int64_t *)seg_key, n_cell);
  const int64_t n = length_bound > n_cell + 1 ? length_bound : n_cell + 1;
  hipLaunchKernelGGL(k_resort_commit, dim3(grid_for(n)), dim3(SDM_BLOCK), 0, s, plan, idx,
                     (const int64_t *)out, cell_start, (const int64_t *)cs_new, n_cell, fctl);
  LAUNCH_CHECK();
  return SDM_OK;
}

size_t sdm_compact_scratch(int64_t n) {
  return carve_size(sizeof(int32_t) * COMPACT_WAVES) + carve_size(sizeof(int32_t) * n) +
         carve_size(sizeof(int64_t) * n);
}

extern "C" int sdm_remove_zero_n_or_flagged(sdm_ctx *ctx, const int64_t *multiplicity,
                                            int64_t *idx, int64_t length, int64_t idx_len,
                                            int64_t *new_length) {
  ARG_TRY(ctx && new_length && length >= 0 && length <= idx_len && idx_len < INT32_MAX);
  if (length == 0) { *new_length = 0; return SDM_OK; }
  ARG_TRY(multiplicity && idx);
  int rc = sdm_reserve(ctx, sdm_compact_scratch(length) + 512);
  if (rc) return rc;
  Carver cv(ctx->arena);
  int64_t *fctl = cv.take<int64_t>(8);   // {valid, work, sorted, healthy = 0: "compact now", ..}
  int64_t *cctl = cv.take<int64_t>(8);
  const int64_t words[8] = {length, length, 0, 0, 0, 0, 0, 0};
  memcpy(ctx->mailbox, words, sizeof(words));
  HIP_TRY(hipMemcpyAsync(fctl, ctx->mailbox, sizeof(words), hipMemcpyHostToDevice, ctx->stream));
  rc = sdm_compact_fused_async(ctx, ctx->arena + 512, multiplicity, idx, length, idx_len, fctl,
                               cctl, nullptr);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, fctl, sizeof(words), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->mailbox[7] != 0) {
    (void)sdm_compact_rearm(ctx);
    sdm_set_error("remove_zero_n_or_flagged: grid barrier of the compaction kernel timed out");
    return SDM_E_HIP;
  }
  *new_length = ctx->mailbox[FCTL_VALID];
  return SDM_OK;
}

// ---------------------------------------------------------------------------------------
// counting sort by cell (collisions_methods.py:682-697): stable sort of idx[0:length) by
// key = cell_idx[cell_id[idx[i]]]; cell_start = exclusive prefix of the key histogram.
// One wavefront per tile: per-tile histograms -> column scan -> in-order scatter where equal
// keys inside a 64-chunk are ranked with ballot-built peer masks.
// ---------------------------------------------------------------------------------------
#define SORT_TILE 1024

__device__ __forceinline__ int64_t sort_key(const int64_t *__restrict__ idx,
                                            const int64_t *__restrict__ cell_id,
                                            const int64_t *__restrict__ cell_idx, int64_t i) {
  return cell_idx[cell_id[idx[i]]];
}

// every kernel of the sort exits at once when the length word is 0 (the fused step gates the
// whole sort on the device that way)
__global__ void __launch_bounds__(SDM_BLOCK)
k_sort_zero(int32_t *__restrict__ H, int64_t n, const int64_t *__restrict__ p_length) {
  if (*p_length == 0) return;
  for (int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * SDM_BLOCK)
    H[i] = 0;
}

// H[key][tile]: per-tile histogram of the keys; keys cached for the scatter pass.  Equal keys
// inside a 64-chunk share one atomic (input that is already grouped by cell -- every re-sort
// after the first -- then needs one atomic per chunk).  One wavefront walks a tile; the dependent
// look-ups idx -> cell_id -> cell_idx of SORT_UNROLL chunks are in flight together (one chunk at a
// time the walk is latency-bound: 76 us at 2^22 super-droplets)
#define SORT_UNROLL 4
__device__ __forceinline__ unsigned long long same_key_lanes(int32_t key, bool in, int key_bits) {
  unsigned long long peers = __ballot(in);
  for (int b = 0; b < key_bits; ++b) {
    const bool bit = (key >> b) & 1;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

__global__ void __launch_bounds__(SDM_WAVE)
k_sort_hist(int32_t *__restrict__ H, int32_t *__restrict__ keys, const int64_t *__restrict__ idx,
            const int64_t *__restrict__ cell_id, const int64_t *__restrict__ cell_idx,
            const int64_t *__restrict__ p_length, int nb, int64_t tile, int key_bits) {
  const int64_t length = *p_length;
  const int64_t first = (int64_t)blockIdx.x * tile;
  const int lane = threadIdx.x;
  for (int64_t base = first; base < first + tile && base < length;
       base += SDM_WAVE * SORT_UNROLL) {
    bool in[SORT_UNROLL];
    int64_t v[SORT_UNROLL];
    int32_t key[SORT_UNROLL];
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; ++u) {
      const int64_t i = base + u * SDM_WAVE + lane;
      in[u] = i < first + tile && i < length;
      v[u] = in[u] ? idx[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; ++u) v[u] = in[u] ? cell_id[v[u]] : 0;
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; ++u) {
      key[u] = in[u] ? (int32_t)cell_idx[v[u]] : -1;
      if (in[u]) keys[base + u * SDM_WAVE + lane] = key[u];
    }
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; ++u) {
      const unsigned long long peers = same_key_lanes(key[u], in[u], key_bits);
      if (in[u] && lane == 63 - __clzll(peers))
        atomicAdd(&H[(int64_t)key[u] * nb + blockIdx.x], __popcll(peers));
    }
  }
}

// one workgroup per key: exclusive scan along the tiles; total -> count[key]
__global__ void __launch_bounds__(BIN_THREADS)
k_sort_colscan(int32_t *__restrict__ H, int64_t *__restrict__ count, int nb,
               const int64_t *__restrict__ p_length) {
  if (*p_length == 0) return;
  int32_t *row = H + (int64_t)blockIdx.x * nb;
  const int per = (nb + BIN_THREADS - 1) / BIN_THREADS;
  const int t0 = threadIdx.x * per;
  int sum = 0;
  for (int k = 0; k < per; ++k)
    if (t0 + k < nb) sum += row[t0 + k];
  int all;
  int run = block_excl_scan(sum, &all);
  for (int k = 0; k < per; ++k)
    if (t0 + k < nb) {
      const int v = row[t0 + k];
      row[t0 + k] = run;
      run += v;
    }
  if (threadIdx.x == 0) count[blockIdx.x] = all;
}

// cell_start[0..n_cell] = exclusive scan of count (single block, chunked)
__global__ void __launch_bounds__(1024)
k_sort_cellstart(const int64_t *__restrict__ count, int64_t *__restrict__ cell_start,
                 int64_t n_cell, const int64_t *__restrict__ p_length) {
  if (*p_length == 0) return;
  __shared__ int64_t sm[1024];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_cell; base += 1024) {
    const int64_t c = base + threadIdx.x;
    const int64_t v = c < n_cell ? count[c] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int64_t t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    const int64_t incl = sm[threadIdx.x];
    if (c < n_cell) cell_start[c] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) cell_start[n_cell] = carry;
}

// stable: the tile's wavefront takes its chunks in order; a group of equal keys inside a chunk
// gets its base from the tile's running counter H[key][tile] (exclusive prefix over the tiles
// after the column scan).  The atomics of SORT_UNROLL chunks are issued back to back - requests of
// one wavefront to one address are served in issue order - and their results used afterwards.
int sdm_cell_start_from_counts_async(sdm_ctx *ctx, const int64_t *count, int64_t *cell_start,
                                     int64_t n_cell, const int64_t *p_gate) {
  hipLaunchKernelGGL(k_sort_cellstart, dim3(1), dim3(1024), 0, ctx->stream, count, cell_start,
                     n_cell, p_gate);
  LAUNCH_CHECK();
  return SDM_OK;
}

__global__ void __launch_bounds__(SDM_WAVE)
k_sort_scatter(int64_t *__restrict__ new_idx, int32_t *__restrict__ H,
               const int32_t *__restrict__ keys, const int64_t *__restrict__ idx,
               const int64_t *__restrict__ cell_start, const int64_t *__restrict__ p_length,
               int nb, int64_t tile, int key_bits) {
  const int64_t length = *p_length;
  const int64_t first = (int64_t)blockIdx.x * tile;
  const int lane = threadIdx.x;
  for (int64_t base = first; base < first + tile && base < length;
       base += SDM_WAVE * SORT_UNROLL) {
    bool in[SORT_UNROLL];
    int64_t v[SORT_UNROLL], start[SORT_UNROLL];
    int32_t key[SORT_UNROLL], basepos[SORT_UNROLL];
    int rank[SORT_UNROLL], leader[SORT_UNROLL];
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; ++u) {
      const int64_t i = base + u * SDM_WAVE + lane;
      in[u] = i < first + tile && i < length;
      v[u] = in[u] ? idx[i] : 0;
      key[u] = in[u] ? keys[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; ++u) {
      start[u] = in[u] ? cell_start[key[u]] : 0;
      const unsigned long long peers = same_key_lanes(key[u], in[u], key_bits);
      rank[u] = __popcll(peers & ((1ull << lane) - 1));
      leader[u] = in[u] ? 63 - __clzll(peers) : 0;
      basepos[u] = 0;
      if (in[u] && lane == leader[u])
        basepos[u] = atomicAdd(&H[(int64_t)key[u] * nb + blockIdx.x], __popcll(peers));
    }
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; ++u) {
      const int32_t b = __shfl(basepos[u], leader[u], 64);
      if (in[u]) new_idx[start[u] + b + rank[u]] = v[u];
    }
  }
}

__global__ void __launch_bounds__(SDM_BLOCK)
k_sort_single_cell(int64_t *__restrict__ new_idx, const int64_t *__restrict__ idx,
                   const int64_t *__restrict__ p_length, int64_t *__restrict__ cell_start) {
  const int64_t length = *p_length;
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i < length) new_idx[i] = idx[i];
  if (i == 0) { cell_start[0] = 0; cell_start[1] = length; }
}

static void sort_geometry(int64_t length_bound, int64_t n_cell, int64_t *tile, int *nb) {
  // keep the tile-histogram matrix below 256 MiB
  int64_t t = SORT_TILE;
  const int64_t max_rows = (int64_t)(64 << 20) / (n_cell > 0 ? n_cell : 1);
  while ((length_bound + t - 1) / t > (max_rows > 1 ? max_rows : 1)) t *= 2;
  *tile = t;
  *nb = (int)((length_bound + t - 1) / t);
  if (*nb < 1) *nb = 1;
}

size_t sdm_sort_scratch(int64_t length_bound, int64_t n_cell) {
  if (n_cell <= 1) return 256;
  int64_t tile;
  int nb;
  sort_geometry(length_bound, n_cell, &tile, &nb);
  return carve_size(sizeof(int32_t) * (size_t)nb * n_cell) + carve_size(sizeof(int64_t) * n_cell) +
         carve_size(sizeof(int32_t) * length_bound);
}

int sdm_counting_sort_async(sdm_ctx *ctx, char *scratch, int64_t *new_idx, const int64_t *idx,
                            const int64_t *cell_id, const int64_t *cell_idx,
                            const int64_t *p_length, int64_t length_bound, int64_t *cell_start,
                            int64_t n_cell) {
  if (n_cell == 1) {
    hipLaunchKernelGGL(k_sort_single_cell, dim3(grid_for(length_bound)), dim3(SDM_BLOCK), 0,
                       ctx->stream, new_idx, idx, p_length, cell_start);
    LAUNCH_CHECK();
    return SDM_OK;
  }
  int64_t tile;
  int nb;
  sort_geometry(length_bound, n_cell, &tile, &nb);
  Carver cv(scratch);
  int32_t *H = cv.take<int32_t>((size_t)nb * n_cell);
  int64_t *count = cv.take<int64_t>(n_cell);
  int32_t *keys = cv.take<int32_t>(length_bound);
  int key_bits = 1;
  while ((1ll << key_bits) < n_cell) ++key_bits;
  hipLaunchKernelGGL(k_sort_zero, dim3(1024), dim3(SDM_BLOCK), 0, ctx->stream, H,
                     (int64_t)nb * n_cell, p_length);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(SDM_WAVE), 0, ctx->stream, H, keys, idx, cell_id,
                     cell_idx, p_length, nb, tile, key_bits);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sort_colscan, dim3((unsigned)n_cell), dim3(BIN_THREADS), 0, ctx->stream, H,
                     count, nb, p_length);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sort_cellstart, dim3(1), dim3(1024), 0, ctx->stream, count, cell_start,
                     n_cell, p_length);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(SDM_WAVE), 0, ctx->stream, new_idx, H, keys,
                     idx, cell_start, p_length, nb, tile, key_bits);
  LAUNCH_CHECK();
  return SDM_OK;
}

extern "C" int sdm_counting_sort_by_cell_id(sdm_ctx *ctx, int64_t *new_idx, const int64_t *idx,
                                            const int64_t *cell_id, const int64_t *cell_idx,
                                            int64_t length, int64_t *cell_start,
                                            int64_t n_cell) {
  ARG_TRY(ctx && cell_start && n_cell >= 1 && length >= 0);
  ARG_TRY(length == 0 || (new_idx && idx && cell_id && cell_idx));
  if (length == 0) {  // nothing to sort: every cell empty
    HIP_TRY(hipMemsetAsync(cell_start, 0, sizeof(int64_t) * (n_cell + 1), ctx->stream));
    return SDM_OK;
  }
  int rc = sdm_reserve(ctx, sdm_sort_scratch(length, n_cell));
  if (rc) return rc;
  int64_t *p_length = ctx->dscal + 8;
  HIP_TRY(hipMemcpyAsync(p_length, &length, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  return sdm_counting_sort_async(ctx, ctx->arena, new_idx, idx, cell_id, cell_idx, p_length,
                                 length > 0 ? length : 1, cell_start, n_cell);
}

// ---- sanitize + re-sort of a sorted state in one call (sdm_hip.h: sdm_sanitize_sorted) --------
// What the fused multi-cell step does after a sub-step in which super-droplets died, as a call of
// its own: the compaction, then the closed-form re-sort or the counting sort.  Lets a caller (and
// the tests) run both re-sorts on one input.
__global__ void k_sanitize_copy_back(int64_t *__restrict__ idx, const int64_t *__restrict__ sorted,
                                     const int64_t *__restrict__ p_length) {
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i < *p_length) idx[i] = sorted[i];
}

extern "C" int sdm_sanitize_sorted(sdm_ctx *ctx, const int64_t *multiplicity, int64_t *idx,
                                   int64_t *tmp_idx, int64_t length, int64_t idx_len,
                                   const int64_t *cell_id, const int64_t *cell_idx,
                                   int64_t *cell_start, int64_t n_cell, int resort,
                                   int64_t *new_length, int *path) {
  ARG_TRY(ctx && new_length && length >= 0 && length <= idx_len && idx_len < INT32_MAX &&
          n_cell >= 1 && cell_start);
  ARG_TRY(resort == SDM_RESORT_AUTO || resort == SDM_RESORT_COUNTING_SORT ||
          resort == SDM_RESORT_ALWAYS_ASK);
  if (path) *path = 0;
  if (length == 0) { *new_length = 0; return SDM_OK; }
  ARG_TRY(multiplicity && idx && tmp_idx && cell_id && cell_idx);
  const size_t head = 1024, seg = carve_size(sizeof(int64_t) * (size_t)(n_cell + 1));
  const size_t compact = carve_size(sdm_compact_scratch(length));
  int rc = sdm_reserve(ctx, head + 3 * seg + compact + carve_size(sdm_sort_scratch(length, n_cell)));
  if (rc) return rc;
  Carver cv(ctx->arena);
  int64_t *fctl = cv.take<int64_t>(8);  // {valid, work, sorted, healthy = 0: "compact now", ..}
  int64_t *cctl = cv.take<int64_t>(8);
  int64_t *plan = cv.take<int64_t>(8);
  int64_t *seg_size = (int64_t *)(ctx->arena + head), *seg_key = (int64_t *)(ctx->arena + head + seg);
  int64_t *cs_new = (int64_t *)(ctx->arena + head + 2 * seg);
  char *compact_scratch = ctx->arena + head + 3 * seg, *sort_scratch = compact_scratch + compact;
  hipStream_t s = ctx->stream;
  const int64_t words[8] = {length, length, 1, 0, 0, 0, 0, 0};
  memcpy(ctx->mailbox, words, sizeof(words));
  HIP_TRY(hipMemcpyAsync(fctl, ctx->mailbox, sizeof(words), hipMemcpyHostToDevice, s));
  rc = sdm_compact_fused_async(ctx, compact_scratch, multiplicity, idx, length, idx_len, fctl,
                               cctl, nullptr);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->mailbox, fctl, sizeof(words), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (ctx->mailbox[7] != 0) {
    (void)sdm_compact_rearm(ctx);
    sdm_set_error("sdm_sanitize_sorted: grid barrier of the compaction kernel timed out");
    return SDM_E_HIP;
  }
  const int64_t valid = ctx->mailbox[FCTL_VALID];
  *new_length = valid;
  if (valid == length) return SDM_OK;  // nothing was removed: still sorted, cell_start stands
  bool closed_form = false;
  if (resort != SDM_RESORT_COUNTING_SORT) {
    rc = sdm_resort_plan(ctx, compact_scratch, length, cctl, fctl, cell_start, n_cell, plan,
                         &closed_form);
    if (rc) return rc;
  }
  if (closed_form) {
    rc = sdm_resort_after_compaction_async(ctx, compact_scratch, length, fctl, idx, tmp_idx,
                                           cell_start, cs_new, cell_id, cell_idx, n_cell, plan,
                                           seg_size, seg_key);
    if (rc) return rc;
    ++ctx->stats[SDM_STAT_RESORT_CLOSED_FORM];
  } else {
    if (resort != SDM_RESORT_COUNTING_SORT) ++ctx->stats[SDM_STAT_RESORT_REFUSED];
    ++ctx->stats[SDM_STAT_RESORT_COUNTING_SORT];
    if (valid == 0) {
      HIP_TRY(hipMemsetAsync(cell_start, 0, sizeof(int64_t) * (size_t)(n_cell + 1), s));
    } else {
      rc = sdm_counting_sort_async(ctx, sort_scratch, tmp_idx, idx, cell_id, cell_idx,
                                   fctl + FCTL_VALID, length, cell_start, n_cell);
      if (rc) return rc;
      hipLaunchKernelGGL(k_sanitize_copy_back, dim3(grid_for(length)), dim3(SDM_BLOCK), 0, s, idx,
                         (const int64_t *)tmp_idx, (const int64_t *)(fctl + FCTL_VALID));
      LAUNCH_CHECK();
    }
  }
  if (path) *path = closed_form ? 2 : 1;
  HIP_TRY(hipStreamSynchronize(s));
  return SDM_OK;
}

// ---------------------------------------------------------------------------------------
// cell_id = strides . cell_origin  (collisions_methods.py:407-416)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SDM_BLOCK)
k_cell_id(int64_t *__restrict__ cell_id, const int64_t *__restrict__ cell_origin,
          const int64_t *__restrict__ strides, int64_t n_dim, int64_t n_sd) {
  const int64_t i = (int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x;
  if (i >= n_sd) return;
  int64_t s = 0;
  for (int64_t d = 0; d < n_dim; ++d) s += strides[d] * cell_origin[d * n_sd + i];
  cell_id[i] = s;
}

extern "C" int sdm_cell_id(sdm_ctx *ctx, int64_t *cell_id, const int64_t *cell_origin,
                           const int64_t *strides, int64_t n_dim, int64_t n_sd) {
  ARG_TRY(ctx && n_sd >= 0 && n_dim >= 1 && (n_sd == 0 || (cell_id && cell_origin && strides)));
  if (n_sd == 0) return SDM_OK;
  hipLaunchKernelGGL(k_cell_id, dim3(grid_for(n_sd)), dim3(SDM_BLOCK), 0, ctx->stream, cell_id,
                     cell_origin, strides, n_dim, n_sd);
  LAUNCH_CHECK();
  return SDM_OK;
}
