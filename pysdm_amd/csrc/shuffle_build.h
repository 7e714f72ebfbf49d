// shuffle_build.h -- the tile sort of the binned shuffle build as a device function: used by
// k_bin_sort / k_bin_build2 (index.hip) and by the pair kernel that sorts the NEXT sub-step's events
// on the side (fused.hip: k_pair_all_sort)
#pragma once
#include "common.h"
#include "shuffle_device.h"

#ifdef __HIPCC__
// Tile shapes (overridable at compile time for tuning runs: -DBIN_SHIFT=.. -DEV_TILE=.. ..).
// Measured on MI355X at n_sd = 2^20 (profiles/README.md, "tile shapes"): 256 -> 1024 threads and
// 2048 -> 4096 positions per bin took the build from 42 to 35 us; smaller or larger event tiles,
// 8 PCG64 steps per thread and 8192-position bins were all slower.
#ifndef BIN_SHIFT
#define BIN_SHIFT 12
#endif
#define BIN_POS (1 << BIN_SHIFT)  // positions per bin (K4 workgroup)
#ifndef BIN_THREADS
#define BIN_THREADS 1024
#endif
#ifndef EV_TILE
#define EV_TILE 4096     // events per K1 / K3 workgroup
#endif
// ... and 16384 where that makes a bin's run in a tile a whole 64-byte sector of successor words
// again (2^20 < n_sd <= 2^22: 1024 bins; index.hip, shuffle_binned_async)
#define EV_TILE_BIG 16384
#define EV_PER_THREAD (EV_TILE / BIN_THREADS)  // K3
#ifndef K1_PER_THREAD
#define K1_PER_THREAD 4  // K1: sequential PCG64 steps per thread kept short
#endif
#define K1_THREADS (EV_TILE / K1_PER_THREAD)
static_assert(EV_PER_THREAD * BIN_THREADS == EV_TILE && K1_THREADS * K1_PER_THREAD == EV_TILE &&
              K1_THREADS <= 1024 && BIN_THREADS % 64 == 0, "tile shapes");

// PackRec and the backward walk live in shuffle_device.h (shared with the fused pair kernels)

// own-event targets of BIN_PER_THREAD consecutive positions from `first` (local croupier)
template <bool RNG, int PER>
__device__ __forceinline__ void targets_run(int64_t first, int64_t length,
                                          const double *__restrict__ u01,
                                          const int64_t *__restrict__ cell_start, int64_t n_cell,
                                          u128 s_tile, u128 inc, const u128 *__restrict__ tab,
                                          int32_t (&j)[PER], int64_t one_cell_len = -1,
                                          bool at_run = false) {
  // at_run: s_tile is already the state at this thread's run
  u128 state = 0;
  if (RNG) state = at_run ? s_tile : pcg_jump(s_tile, tab, (uint64_t)threadIdx.x * PER);
  const u128 mult = pcg_mult();
  int64_t lo = 0, hi = 0;
  bool have_cell = false;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int64_t i = first + e;
    double u = 0.0;
    if (RNG) {
      state = state * mult + inc;
      u = pcg_output(state);
    } else if (i < length) {
      u = u01[i];
    }
    j[e] = -1;
    if (i >= length) continue;
    if (!have_cell || i >= hi) {
      if (one_cell_len >= 0) {  // one cell [0, length): known to the caller (see k_bin_sort)
        lo = 0;
        hi = one_cell_len;
      } else {
        const int64_t c = n_cell == 1 ? 0 : find_cell(cell_start, n_cell, i);
        lo = cell_start[c];
        hi = cell_start[c + 1];
      }
      have_cell = true;
    }
    if (i > lo) {
      const int64_t t = (int64_t)((double)lo + u * (double)(hi - lo));
      // memory safety only: the reference would index past the cell with prob ~2^-43
      j[e] = (int32_t)(t > hi - 1 ? hi - 1 : (t < lo ? lo : t));
    }
  }
}

// block-wide exclusive scan of one value per thread (BIN_THREADS threads); returns the exclusive
// prefix, *total receives the block sum
__device__ __forceinline__ int block_excl_scan(int v, int *total) {
  __shared__ int wsum[BIN_THREADS / SDM_WAVE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  __syncthreads();  // protects wsum against the previous call
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0, sum = 0;
#pragma unroll
  for (int k = 0; k < BIN_THREADS / SDM_WAVE; ++k) {
    if (k < w) base += wsum[k];
    sum += wsum[k];
  }
  *total = sum;
  return base + incl - v;
}

// ---- two launches (round 1 had four: count -> column scan -> scatter -> build; the count matrix,
// its scan and the global scatter are gone: 32.5 -> 28.5 us of kernel time at 2^20, two launches
// less).  K1': per event tile: own-event targets -> jarr; the tile's events ordered by target bin in LDS
// and written back *in place* (tile-major, coalesced), with the tile's bin offsets toff[tile][0..nb].
// K4': one workgroup per bin gathers its runs - for every tile the events toff[t][b]..toff[t][b+1]
// of that tile's segment, ~16 events = two 64-B sectors each, the same granularity the scatter
// wrote at - and assembles the records as k_bin_build does.  The order of the events inside a bin
// is irrelevant (the walk takes the minimum over a position's candidates).
// -DBIN_PROFILE (tuning builds only): phase time stamps of workgroup 7 of the two build kernels,
// wall_clock64 ticks (100 MHz), read back through sdm_debug_bin_profile
#ifdef BIN_PROFILE
extern __device__ long long bin_prof[32];
#define BIN_MARK(k) do { __syncthreads(); if (blockIdx.x == 7 && threadIdx.x == 0) bin_prof[k] = wall_clock64(); } while (0)
#else
#define BIN_MARK(k)
#endif

// `one_cell_len` >= 0: the single cell [0, length) as the caller knows it (cell_start not read)
template <bool RNG, int TILE = EV_TILE>
__device__ __forceinline__ void
bin_sort_body(char *smem, int2 *__restrict__ events, int32_t *__restrict__ toff,
              int32_t *__restrict__ jarr, int32_t *__restrict__ loc, int n_bins,
              const double *__restrict__ u01,
              const int64_t *__restrict__ cell_start, int64_t n_cell, int64_t length,
              int64_t one_cell_len, u128 s_off, u128 inc, const u128 *__restrict__ tab,
              const uint64_t *__restrict__ dev_off, const u128 *__restrict__ aff) {
  int32_t *lstart = (int32_t *)smem;                      // n_bins + 1
  int32_t *lcount = lstart + n_bins + 1;                  // n_bins
  int2 *ev_buf = (int2 *)(lcount + ((n_bins + 1) & ~1));  // TILE
  __shared__ u128 s_slot;
  BIN_MARK(0);
  constexpr int PER = TILE / BIN_THREADS;  // consecutive events per thread
  static_assert(PER * BIN_THREADS == TILE && TILE % PCG_AFF_STRIDE == 0, "tile shapes");
  const int64_t tile_first = (int64_t)blockIdx.x * TILE;
  int32_t *my_off = toff + (int64_t)blockIdx.x * (n_bins + 1);
  if (tile_first >= length) {
    for (int b = threadIdx.x; b <= n_bins; b += BIN_THREADS) my_off[b] = 0;
    return;
  }
  for (int b = threadIdx.x; b < n_bins; b += BIN_THREADS) lcount[b] = 0;
  // The jump-aheads (to the tile, then to the thread's run) sit on the critical path of a
  // one-workgroup-per-CU kernel.  Ready affine maps (ctx->pcg_aff) make them two 128-bit
  // multiply-adds per thread and nothing to wait for (4.6 + 1.2 us of this kernel's 12.5 were
  // the bit-by-bit jumps, `profiles/r02_bin_profile.txt`); otherwise bit by bit, the table in LDS
  // (a thread's run starts tid * PER draws into the tile: whole strides of the table + a rest)
  constexpr int STRIDES = TILE / PCG_AFF_STRIDE;
  const bool ready = RNG && aff && !dev_off &&
                     ((int64_t)blockIdx.x + 1) * STRIDES <= PCG_AFF_TILES;
  __shared__ u128 ltab[128];
  u128 s_tile = 0;
  if (ready) {
    const int into = (int)threadIdx.x * PER;
    s_tile = pcg_apply(pcg_apply(s_off, aff, PCG_AFF_SMALL + (int64_t)blockIdx.x * STRIDES +
                                                 into / PCG_AFF_STRIDE),
                       aff, (int64_t)(into % PCG_AFF_STRIDE));
    __syncthreads();  // lcount is zero before anyone counts
  } else if (RNG) {
    if (threadIdx.x < 128) ltab[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    tab = ltab;
    // dev_off (graph replay): s_off is the generator's initial state, the stream position comes
    // from the device
    if (threadIdx.x == 0)
      s_slot = pcg_jump(s_off, tab, (uint64_t)blockIdx.x * TILE + (dev_off ? dev_off[0] : 0));
    __syncthreads();
    s_tile = s_slot;
  } else {
    __syncthreads();  // lcount is zero before anyone counts
  }
  BIN_MARK(1);
  BIN_MARK(2);
  const int64_t first = tile_first + (int64_t)threadIdx.x * PER;
  int32_t j[PER];
  targets_run<RNG, PER>(first, length, u01, cell_start, n_cell, s_tile, inc, tab, j,
                                  one_cell_len, ready);
  BIN_MARK(3);
  int rank[PER];  // arrival number of the event in its bin: its place in the bin's run
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    rank[e] = j[e] >= 0 ? atomicAdd(&lcount[j[e] >> BIN_SHIFT], 1) : 0;
    if (first + e < length) jarr[first + e] = j[e];
  }
  __syncthreads();
  BIN_MARK(4);
  {  // lstart = exclusive scan of lcount
    const int per = (n_bins + BIN_THREADS - 1) / BIN_THREADS;
    const int b0 = threadIdx.x * per;
    int sum = 0;
    for (int k = 0; k < per; ++k)
      if (b0 + k < n_bins) sum += lcount[b0 + k];
    int all;
    int run = block_excl_scan(sum, &all);
    for (int k = 0; k < per; ++k)
      if (b0 + k < n_bins) {
        lstart[b0 + k] = run;
        run += lcount[b0 + k];
      }
    if (threadIdx.x == 0) lstart[n_bins] = all;
  }
  __syncthreads();
  BIN_MARK(5);
#pragma unroll
  for (int e = 0; e < PER; ++e)
    if (j[e] >= 0) {
      const int b = j[e] >> BIN_SHIFT;
      const int at = lstart[b] + rank[e];
      ev_buf[at] = make_int2((int)(first + e), j[e]);
      // (SDM_REC_CHAIN: where the event stands in the sorted array - its S word's address)
      if (loc) loc[first + e] = (int32_t)(tile_first + at);
    }
  __syncthreads();
  BIN_MARK(6);
  const int n_ev = lstart[n_bins];
  for (int t = threadIdx.x; t < n_ev; t += BIN_THREADS) events[tile_first + t] = ev_buf[t];
  for (int b = threadIdx.x; b <= n_bins; b += BIN_THREADS) my_off[b] = lstart[b];
  BIN_MARK(7);
}

#endif  // __HIPCC__
