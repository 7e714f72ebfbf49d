// index.h -- async cores of the permutation-side kernels, shared with the fused step
#pragma once
#include "common.h"
#include "shuffle_device.h"

int sdm_pcg_prepare(sdm_ctx *ctx, const uint64_t state_inc[4]);
u128 sdm_pcg_advance_host(u128 state, u128 inc, uint64_t delta);
int sdm_pcg_fill_async(sdm_ctx *ctx, double *out, int64_t n, const uint64_t state_inc[4],
                       uint64_t offset);

size_t sdm_shuffle_scratch(int64_t n);
int sdm_shuffle_async(sdm_ctx *ctx, char *scratch, int64_t *out, const int64_t *idx0,
                      const double *u01, const int64_t *cell_start, int64_t n_cell,
                      const int64_t *p_length, int64_t length_bound, bool global,
                      int64_t n_total, const uint64_t *rng_state_inc, uint64_t rng_offset);
int sdm_sort_by_key_async(sdm_ctx *ctx, int64_t *idx, const double *keys, int64_t n);
size_t sdm_compact_scratch(int64_t n);
int sdm_compact_rearm(sdm_ctx *ctx);  // clears the grid-barrier words after a time-out
size_t sdm_sort_scratch(int64_t length_bound, int64_t n_cell);
int sdm_counting_sort_async(sdm_ctx *ctx, char *scratch, int64_t *new_idx, const int64_t *idx,
                            const int64_t *cell_id, const int64_t *cell_idx,
                            const int64_t *p_length, int64_t length_bound, int64_t *cell_start,
                            int64_t n_cell);
// cell_start[0..n_cell] = exclusive scan of count[0..n_cell) (no-op while *p_gate == 0)
int sdm_cell_start_from_counts_async(sdm_ctx *ctx, const int64_t *count, int64_t *cell_start,
                                     int64_t n_cell, const int64_t *p_gate);
bool sdm_shuffle_can_split(int64_t n, bool global);
// The arguments of a compaction run by another kernel of the build (k_bin_build2's prologue)
struct SortPrologue {
  int64_t *fctl;  // NULL: no prologue
  const int64_t *multiplicity;
  int64_t *idx;
  int64_t flag;
  int32_t *wave_dead;
  int n_tiles;
  int64_t *ctl;
  int32_t *holes;
  int64_t *fillers;
  int64_t *cell_start_single;
  unsigned int *bar;
};
// k_bin_build2 after a tile sort done ahead of time (fused.hip: k_pair_all_sort): the compaction and
// the re-sort it has to do itself when a super-droplet died in between
struct BuildPrologue {
  SortPrologue compact;  // compact.fctl == NULL: none
  // SDM_REC_CHAIN: the overflow links of the hit lists (scratch of the build); per event its place
  // in the sorted array (from the tile sort); the S words, by that place
  int32_t *chain_links, *loc;
  uint32_t *ssucc;
  int2 *events;
  int32_t *toff, *jarr;
  u128 s_off, inc;
  const u128 *tab, *aff;
};
// the buffers of a build, for a tile sort done elsewhere (same scratch, same size)
struct SortBuffers {
  int2 *events;
  int32_t *toff, *jarr, *loc;  // (loc: NULL unless the build makes successor words)
  int n_bins, n_tiles;
  size_t lds_bytes;
};
// true: builds over `length_bound` positions can take their tile sort from the previous pair kernel
// (as many tiles as bins, all workgroups of the build resident at once for the rare re-sort)
bool sdm_shuffle_presort_ok(sdm_ctx *ctx, int64_t length_bound, int64_t id_bound);
void sdm_shuffle_sort_buffers(sdm_ctx *ctx, char *scratch, int64_t length_bound, SortBuffers *out);
void sdm_compact_as_prologue(sdm_ctx *ctx, char *scratch, const int64_t *multiplicity,
                             int64_t *idx, int64_t length_bound, int64_t flag, int64_t *fctl,
                             int64_t *ctl, int64_t *cell_start_single, SortPrologue *out);
int sdm_shuffle_build_async(sdm_ctx *ctx, char *scratch, const int64_t *idx0,
                            const int64_t *cell_start, int64_t n_cell, const int64_t *p_length,
                            int64_t length_bound, const uint64_t *rng_state_inc,
                            uint64_t rng_offset, ShuffleViews *views,
                            int64_t id_bound = -1,  // ids in idx0 are below it (-1: unknown)
                            const uint64_t *dev_off = nullptr,
                            // the tile sort was done ahead (k_pair_all_sort): what the build needs
                            // for the compaction it may have to run first
                            const SortPrologue *presorted = nullptr);
// What ends an adaptive sub-step of a single cell (collision.py:185-187), done by the compaction
// kernel's last act instead of a launch of its own: refused-breakup count of the counter slots
// into fctl[4] (slots may be NULL), working length = dt_left[0] != 0 ? valid length : 0, control
// block published to the polled host box with sequence number `seq` (common.h:publish_ctl)
struct CompactEpilogue {
  const double *dt_left;  // NULL: no epilogue
  int64_t *slots;
  int64_t *box;
  int64_t seq;
  // graph replay (common.h: gwords): stream positions advanced by these at the kernel's start
  uint64_t *gwords;
  uint64_t advance, advance_b;
  // the positions of this sub-step's dead as a list (fused.hip: flag_dead appends them): up to
  // COMPACT_WAVES of them are removed by ONE workgroup without a pass over the permutation
  // (compact_listed_body).  `dead_count` is read-only in the kernel - every workgroup must see the
  // same number - and the counter of the NEXT sub-step is cleared instead (two counters, used in
  // turn).  NULL: no list
  const int64_t *dead_pos;
  const unsigned long long *dead_count;
  unsigned long long *dead_count_next;
  // a copy of dt_left[0] for the next sub-step's k_pair_update (fused.hip: FusedArgs::dt_left_pub)
  double *dt_left_pub;
};
int sdm_resort_plan(sdm_ctx *ctx, char *scratch, int64_t length_bound, const int64_t *cctl,
                    const int64_t *fctl, const int64_t *cell_start, int64_t n_cell, int64_t *plan,
                    bool *applies);
int sdm_resort_after_compaction_async(sdm_ctx *ctx, char *scratch, int64_t length_bound,
                                      int64_t *fctl, int64_t *idx, int64_t *out,
                                      int64_t *cell_start, int64_t *cs_new,
                                      const int64_t *cell_id, const int64_t *cell_idx,
                                      int64_t n_cell, const int64_t *plan, int64_t *seg_size,
                                      int64_t *seg_key);
// the same compaction from a device list of the d positions to remove (all flagged already);
// d within sdm_compact_listed_fits, else the full kernel
bool sdm_compact_listed_fits(int64_t d);
int sdm_compact_listed_async(sdm_ctx *ctx, char *scratch, int64_t *idx, const int64_t *dead,
                             int64_t d, int64_t length_bound, int64_t flag, int64_t *fctl,
                             int64_t *ctl, int64_t *cell_start_single);
int sdm_compact_fused_async(sdm_ctx *ctx, char *scratch, const int64_t *multiplicity,
                            int64_t *idx, int64_t length_bound, int64_t flag, int64_t *fctl,
                            int64_t *ctl, int64_t *cell_start_single, bool flag_only = false,
                            const CompactEpilogue *epilogue = nullptr);
