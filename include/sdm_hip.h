/*
 * sdm_hip.h -- C ABI of libsdm_hip.so: the MI355X (gfx950) implementation of the SDM
 * collision / coalescence / breakup hot path behind PySDM's backend-method interface.
 *
 * Every entry point is what a binding of the reference's backend for this path calls: one symbol
 * per reference backend method (cited as reference file:line, relative to the reference root),
 * plus the fused per-time-step entry `sdm_collision_step`.  Plain pointers and sizes only; all
 * array pointers are DEVICE pointers owned by the caller (Storage.INT = int64, Storage.FLOAT =
 * double, Storage.BOOL = uint8, PySDM/backends/impl_numba/storage.py:16-19).  The library keeps
 * no persistent memory except the opaque sdm_ctx (scratch arena + stream).
 *
 * Return value: 0 = ok, negative = SDM_E_* (message via sdm_last_error()).  Functions whose
 * reference counterpart returns a scalar write it through an out-pointer and synchronise the
 * ctx stream; all other functions only enqueue work on the ctx stream.
 */
#ifndef SDM_HIP_H
#define SDM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDM_OK 0
#define SDM_E_ARG (-1)    /* bad argument (null pointer, negative size, unsupported option) */
#define SDM_E_HIP (-2)    /* a HIP runtime call failed */
#define SDM_E_NOMEM (-3)  /* scratch arena allocation failed */
#define SDM_E_STATE (-4)  /* the state handed over is inconsistent: a sub-step loop did not end
                             within the bound its own arithmetic sets (see sdm_collision_step) */

typedef struct sdm_ctx sdm_ctx;

/* ---- context ------------------------------------------------------------------------- */
int sdm_ctx_create(sdm_ctx **out, int device);
int sdm_ctx_destroy(sdm_ctx *ctx);
/* hipStream_t of the caller (e.g. torch's current stream); NULL = default stream */
int sdm_ctx_set_stream(sdm_ctx *ctx, void *hip_stream);
int sdm_ctx_synchronize(sdm_ctx *ctx);
const char *sdm_last_error(void);
int sdm_abi_version(void);
/* per-phase timing with HIP events recorded on the ctx stream around the kernels of the fused
 * step (measurement only; cf. the reference's per-dynamic WallTimer, PySDM/impl/wall_timer.py).
 * sdm_ctx_read_timing synchronises, adds the elapsed times of all recorded regions to ms[phase] /
 * count[phase] (arrays of SDM_N_PHASES) and clears the recording.                            */
#define SDM_PHASE_SORT 0
#define SDM_PHASE_RNG 1
#define SDM_PHASE_SHUFFLE_CLEAR 2
#define SDM_PHASE_SHUFFLE_BUILD 3
#define SDM_PHASE_SHUFFLE_TRACE 4
#define SDM_PHASE_TAIL_COPY 5
#define SDM_PHASE_CELLS_PRE 6
#define SDM_PHASE_PAIR_PROB 7
#define SDM_PHASE_CELLS_ADAPTIVE 8
#define SDM_PHASE_PAIR_UPDATE 9
#define SDM_PHASE_SANITIZE 10
#define SDM_PHASE_ADAPTIVE_END 11
#define SDM_N_PHASES 12
int sdm_ctx_set_timing(sdm_ctx *ctx, int enable);
int sdm_ctx_read_timing(sdm_ctx *ctx, double *ms, int64_t *count);
const char *sdm_phase_name(int phase);

/* ---- options and statistics of a context ---------------------------------------------------
 * SDM_OPT_RESORT: how a multi-cell adaptive run re-sorts by cell after a compaction
 * (particle_attributes.py:51-55,67-73: `sanitize` un-sorts, the `cell_start` getter sorts again):
 *   SDM_RESORT_AUTO (default): the closed form of index.hip (sdm_sanitize_sorted below) where the
 *     device says it applies; after a refusal the next 16 compactions OF THE SAME CALL go straight
 *     to the counting sort (asking costs a host round trip);
 *   SDM_RESORT_COUNTING_SORT: always the counting sort;  SDM_RESORT_ALWAYS_ASK: no back-off.
 * The results are identical either way; the switch exists for measurements and tests.          */
#define SDM_OPT_RESORT 0
#define SDM_RESORT_AUTO 0
#define SDM_RESORT_COUNTING_SORT 1
#define SDM_RESORT_ALWAYS_ASK 2
/* SDM_OPT_MAX_SUBSTEPS (tests): a bound on the sub-steps of one adaptive time step BELOW the one
 * the arithmetic sets (see sdm_collision_step), so that the error path - SDM_E_STATE, the control
 * block in the message, the context usable afterwards - can be exercised with a healthy state;
 * 0 (default) = the arithmetic's bound alone                                                    */
#define SDM_OPT_MAX_SUBSTEPS 1
/* SDM_OPT_CELL_SHAPE (measurements, tests): the shape of the per-cell kernel of a multi-cell run
 * with a local croupier.  SDM_CELL_SHAPE_AUTO (default) picks by the size of the largest cell and
 * the number of cells this process computes; the others force one shape wherever its cell-size cap
 * allows it (beyond the cap: the 512-thread shape or the general kernel).  Results are identical in all.       */
#define SDM_OPT_CELL_SHAPE 2
#define SDM_CELL_SHAPE_AUTO 0
#define SDM_CELL_SHAPE_512 1  /* two workgroups of 512 threads per CU, cells <= 5632; small cells
                                 (eight per workgroup): the 704-position variant also where the
                                 384-position one would be taken */
#define SDM_CELL_SHAPE_1024 2 /* one workgroup of 1024 per CU, cells <= 6144: fewer cells than CUs */
#define SDM_CELL_SHAPE_256 3  /* four workgroups of 256 per CU, cells <= 2816 */
/* Three A/B switches (measurements, and tests that want both implementations in one process).
 * Results are identical either way.  The defaults can also be set for a whole process through
 * the environment, read when a context is created: SDM_REC_FORMAT=records, SDM_NO_PRESORT=1,
 * SDM_CELL_COPY=0.
 * SDM_OPT_REC_FORMAT: SDM_REC_FORMAT_AUTO (default) = successor words where they apply (one cell,
 *   the shuffle left to the pair kernels, up to 2^22 positions); SDM_REC_FORMAT_RECORDS = the
 *   packed 16-byte records of rounds 1-3 everywhere.
 * SDM_OPT_NO_PRESORT: 1 = the tile sort of the next step's shuffle build does not ride in the
 *   pair kernel (one cell, non-adaptive runs): k_bin_sort and the compaction as launches of
 *   their own in every step.
 * SDM_OPT_NO_CELL_COPY: 1 = multi-cell runs of three steps or more do not work on the
 *   cell-ordered copy of the state.                                                             */
#define SDM_OPT_REC_FORMAT 3
#define SDM_REC_FORMAT_AUTO 0
#define SDM_REC_FORMAT_RECORDS 1
#define SDM_OPT_NO_PRESORT 4
#define SDM_OPT_NO_CELL_COPY 5
int sdm_ctx_set_option(sdm_ctx *ctx, int option, int64_t value);
/* counters since the context was created / last cleared (host-side bookkeeping of the library):
 * re-sorts after a compaction done by the closed form; asked for and refused by the device; not
 * asked because of the back-off; done by the counting sort (= refused + skipped + switched off);
 * sub-steps executed by the fused entry points; sub-steps launched ahead that fell through      */
#define SDM_STAT_RESORT_CLOSED_FORM 0
#define SDM_STAT_RESORT_REFUSED 1
#define SDM_STAT_RESORT_SKIPPED 2
#define SDM_STAT_RESORT_COUNTING_SORT 3
#define SDM_STAT_SUBSTEPS 4
#define SDM_STAT_SUBSTEPS_TAKEN_BACK 5
#define SDM_STAT_EXCHANGES 6       /* collectives of sharded steps (callback or RCCL) */
#define SDM_STAT_EXCHANGE_BYTES 7  /* payload handed to them */
#define SDM_N_STATS 8
int sdm_ctx_read_stats(sdm_ctx *ctx, int64_t *stats, int clear);

/* measurement only: the ceiling of the path's access pattern on this device.  The single-cell pair
 * kernel is bound by independent random 16-byte reads (shuffle records, {multiplicity, mass}
 * records: one 64-B sector miss each), not by streaming bandwidth.  Times
 * `repetitions` launches of a kernel that does nothing else - n_reads 16-byte reads at hashed
 * indices out of a table of table_records x 16 B held in the ctx scratch arena (record i = {i,
 * 2 i + 1}) - with HIP events on the ctx stream; *checksum = sum over all timed reads of both
 * words mod 2^64 (read k of repetition r goes to record mix64(k ^ (r + 1) * 0x100000001b3) mod
 * table_records, mix64 = the splitmix64 finaliser), so a caller can verify that the bytes were
 * touched.  Synchronises.  (The reference has no counterpart: its timers are wall-clock,
 * PySDM/impl/wall_timer.py.)                                                                  */
int sdm_calib_random_sectors(sdm_ctx *ctx, int64_t table_records, int64_t n_reads,
                             int repetitions, double *ms_per_launch, uint64_t *checksum);
/* the counterpart for stores: n_writes scattered 8-byte stores (same hash) into a table of
 * table_words x 8 B - what delivering results to random positions would cost (used to price
 * re-routings of the shuffle's walks).  Synchronises.                             */
int sdm_calib_random_writes(sdm_ctx *ctx, int64_t table_words, int64_t n_writes, int repetitions,
                            double *ms_per_launch);

/* ---- a-1 RNG: NumPy PCG64 stream, PySDM/backends/impl_numba/random.py:13-19 ------------
 * out[i] = double number (offset + i) of the stream of PCG64 with the given state/inc
 * (state_inc = {state_hi, state_lo, inc_hi, inc_lo} of numpy.random.PCG64(seed).state).      */
int sdm_pcg64_uniform(sdm_ctx *ctx, double *out, int64_t n, const uint64_t state_inc[4],
                      uint64_t offset);

/* ---- a-2/a-3/a-19 index methods, PySDM/backends/impl_numba/methods/index_methods.py ----- */
int sdm_identity_index(sdm_ctx *ctx, int64_t *idx, int64_t n);                   /* :14-20 */
int sdm_shuffle_global(sdm_ctx *ctx, int64_t *idx, int64_t length, const double *u01); /* :22-29 */
int sdm_shuffle_local(sdm_ctx *ctx, int64_t *idx, const double *u01, const int64_t *cell_start,
                      int64_t n_cell);                                           /* :32-43 */
int sdm_sort_by_key(sdm_ctx *ctx, int64_t *idx, const double *keys, int64_t n);  /* :46-48 */

/* ---- a-18 / a-4, PySDM/backends/impl_numba/methods/collisions_methods.py --------------- */
/* :664-680 ; idx_len = len(idx) = the "removed" sentinel; *new_length out (host), syncs */
int sdm_remove_zero_n_or_flagged(sdm_ctx *ctx, const int64_t *multiplicity, int64_t *idx,
                                 int64_t length, int64_t idx_len, int64_t *new_length);
/* `ParticleAttributes.sanitize` (particle_attributes.py:67-73) followed by what the `cell_start`
 * getter then does (:51-55 -> __sort_by_cell_id :106-110 -> the counting sort,
 * collisions_methods.py:587-631,682-697) for a state that IS sorted by cell on entry (cell_start
 * describes idx[0:length)): entries with idx[i] == idx_len (flagged) or zero multiplicity are
 * removed the reference's way (swap from the end), the survivors stably sorted by
 * cell_idx[cell_id[.]] again; idx[0:*new_length) and cell_start are rewritten in place, tmp_idx is
 * scratch.  `resort`: SDM_RESORT_COUNTING_SORT, or SDM_RESORT_AUTO / _ALWAYS_ASK = the closed form
 * where it applies: a sorted state stays grouped by cell under the swap-from-the-end except for
 * the fillers, and if the whole removed tail lay in the last non-empty segment, members keep
 * their order, the fillers of earlier segments' holes line up in front of the last segment in
 * hole order and whole segments move to where the current cell_idx puts them - no histogram over
 * gathered keys.  *path (host, may be NULL): 0 = nothing to remove (state untouched), 1 = counting
 * sort, 2 = closed form.  The result does not depend on the path.  Synchronises.                */
int sdm_sanitize_sorted(sdm_ctx *ctx, const int64_t *multiplicity, int64_t *idx, int64_t *tmp_idx,
                        int64_t length, int64_t idx_len, const int64_t *cell_id,
                        const int64_t *cell_idx, int64_t *cell_start, int64_t n_cell, int resort,
                        int64_t *new_length, int *path);
/* :587-631,682-697 ; writes new_idx[0:length) and cell_start[0:n_cell+1] (caller swaps buffers) */
int sdm_counting_sort_by_cell_id(sdm_ctx *ctx, int64_t *new_idx, const int64_t *idx,
                                 const int64_t *cell_id, const int64_t *cell_idx, int64_t length,
                                 int64_t *cell_start, int64_t n_cell);
/* :407-416 ; cell_origin is (n_dim, n_sd) row-major */
int sdm_cell_id(sdm_ctx *ctx, int64_t *cell_id, const int64_t *cell_origin,
                const int64_t *strides, int64_t n_dim, int64_t n_sd);

/* ---- a-5..a-7 pair methods, PySDM/backends/impl_numba/methods/pair_methods.py ---------- */
int sdm_find_pairs(sdm_ctx *ctx, const int64_t *cell_start, uint8_t *is_first_in_pair,
                   const int64_t *cell_id, const int64_t *cell_idx, const int64_t *idx,
                   int64_t length);                                              /* :34-55 */
/* :126-140 ; attr_is_int: 1 = int64 column (multiplicity), 0 = double column */
int sdm_sort_within_pair_by_attr(sdm_ctx *ctx, int64_t *idx, int64_t length,
                                 const uint8_t *is_first_in_pair, const void *attr,
                                 int attr_is_int);
/* op: 0 sum :142-160, 1 max :57-75, 2 min :77-95, 3 distance :14-32, 4 multiply :162-180.
 * Zero-fills out[0:n_out) first (semantic: prob == 0 means "no pair").                      */
#define SDM_PAIR_SUM 0
#define SDM_PAIR_MAX 1
#define SDM_PAIR_MIN 2
#define SDM_PAIR_DISTANCE 3
#define SDM_PAIR_MULTIPLY 4
int sdm_pair_op(sdm_ctx *ctx, int op, double *out, int64_t n_out, const void *in, int in_is_int,
                const uint8_t *is_first_in_pair, const int64_t *idx, int64_t length);
int sdm_sort_pair(sdm_ctx *ctx, double *out, int64_t n_out, const double *in,
                  const uint8_t *is_first_in_pair, const int64_t *idx, int64_t length); /* :97-124 */

/* ---- a-10..a-14 collisions methods, collisions_methods.py ------------------------------ */
int sdm_normalize(sdm_ctx *ctx, double *prob, int64_t n_prob, const int64_t *cell_id,
                  const int64_t *cell_idx, const int64_t *cell_start, double *norm_factor,
                  int64_t n_cell, double timestep, double dv);                   /* :633-662 */
int sdm_scale_prob_for_adaptive_sdm_gamma(sdm_ctx *ctx, double *prob, const int64_t *idx,
                                          int64_t length, const int64_t *multiplicity,
                                          const int64_t *cell_id, double *dt_left, int64_t n_cell,
                                          double dt, double dt_min, double dt_max,
                                          const uint8_t *is_first_in_pair,
                                          int64_t *stats_n_substep,
                                          double *stats_dt_min);                 /* :330-405 */
int sdm_compute_gamma(sdm_ctx *ctx, const double *prob, const double *rand, const int64_t *idx,
                      int64_t length, const int64_t *multiplicity, const int64_t *cell_id,
                      int64_t *collision_rate_deficit, int64_t *collision_rate,
                      const uint8_t *is_first_in_pair, double *out);             /* :522-585 */
/* :313-328 ; *end out (host), syncs */
int sdm_adaptive_sdm_end(sdm_ctx *ctx, const double *dt_left, int64_t n_cell,
                         const int64_t *cell_start, int64_t *end);
/* :418-453 (+ coalesce :44-59, flag_zero_multiplicity :38-41); attributes (n_attr, n_sd) */
int sdm_collision_coalescence(sdm_ctx *ctx, int64_t *multiplicity, const int64_t *idx,
                              int64_t length, double *attributes, int64_t n_attr, int64_t n_sd,
                              const double *gamma, int64_t *healthy, const int64_t *cell_id,
                              int64_t *coalescence_rate, const uint8_t *is_first_in_pair);
/* :247-311 (+ :62-243); *n_overflow (device int64, may be NULL) counts "overflow" warnings */
int sdm_collision_coalescence_breakup(
    sdm_ctx *ctx, int64_t *multiplicity, const int64_t *idx, int64_t length, double *attributes,
    int64_t n_attr, int64_t n_sd, const double *gamma, const double *rand, const double *Ec,
    const double *Eb, const double *fragment_mass, int64_t *healthy, const int64_t *cell_id,
    int64_t *coalescence_rate, int64_t *breakup_rate, int64_t *breakup_rate_deficit,
    const uint8_t *is_first_in_pair, int64_t max_multiplicity, const double *particle_mass,
    int handle_all_breakups, int64_t *n_overflow);
/* :743-782 ; params[13] host array */
int sdm_linear_collection_efficiency(sdm_ctx *ctx, const double params[13], double *output,
                                     int64_t n_out, const double *radii,
                                     const uint8_t *is_first_in_pair, const int64_t *idx,
                                     int64_t length, double unit);

/* ---- a-9 derived attributes ------------------------------------------------------------- */
/* PySDM/backends/impl_numba/methods/terminal_velocity_methods.py:14-30 */
int sdm_interpolation(sdm_ctx *ctx, double *output, const double *radius, int64_t n,
                      double factor, const double *b, const double *c, int64_t table_len);
/* PySDM/backends/impl_numba/methods/physics_methods.py:107-131 (liquid spheres) */
int sdm_volume_of_water_mass(sdm_ctx *ctx, double *volume, const double *mass, int64_t n,
                             double rho_w);
int sdm_mass_of_water_volume(sdm_ctx *ctx, double *mass, const double *volume, int64_t n,
                             double rho_w);

/* ---- a-17 fragmentation, PySDM/backends/impl_numba/methods/fragmentation_methods.py ----- */
/* :136-171 exp_fragmentation incl. limiters :76-95 ; nfmax < 0 == None */
int sdm_exp_fragmentation(sdm_ctx *ctx, double *n_fragment, double scale, double *frag_volume,
                          const double *x_plus_y, const double *rand, int64_t n, double vmin,
                          double nfmax, double tol);
/* :218-257,321-377 straub_fragmentation incl. limiters; consts = {CM, STRAUB_E_D1, STRAUB_MU2,
 * VEDDER_1987_A, VEDDER_1987_b, PI} (host array)                                            */
int sdm_straub_fragmentation(sdm_ctx *ctx, double *n_fragment, const double *CW,
                             const double *gam, const double *ds, double *frag_volume,
                             const double *v_max, const double *x_plus_y, const double *rand,
                             int64_t n, double vmin, double nfmax, double *Nr1, double *Nr2,
                             double *Nr3, double *Nr4, double *Nrt, double *d34,
                             const double consts[6]);

/* f-2 rows: :477-485 gauss (consts = {VEDDER_1987_A, VEDDER_1987_b}), :487-499 feingold1988,
 * :98-112 slams -- each followed by the limiters :76-95                                       */
int sdm_gauss_fragmentation(sdm_ctx *ctx, double *n_fragment, double mu, double sigma,
                            double *frag_volume, const double *x_plus_y, const double *rand,
                            int64_t n, double vmin, double nfmax, const double consts[2]);
int sdm_feingold1988_fragmentation(sdm_ctx *ctx, double *n_fragment, double scale,
                                   double *frag_volume, const double *x_plus_y,
                                   const double *rand, int64_t n, double fragtol, double vmin,
                                   double nfmax);
int sdm_slams_fragmentation(sdm_ctx *ctx, double *n_fragment, double *frag_volume,
                            const double *x_plus_y, double *probs, const double *rand, int64_t n,
                            double vmin, double nfmax);

/* fragmentation_methods.py:260-319,379-474 (Low & List 1982): frag_volume per pair from the
 * filament / sheet / disk modes; rand, Rf, Rs, Rd are updated in place as in the reference;
 * consts = {CM, PI, VEDDER_1987_A, VEDDER_1987_b}; limiters (:75-93) applied afterwards.      */
int sdm_ll82_fragmentation(sdm_ctx *ctx, double *n_fragment, const double *CKE, const double *W,
                           const double *W2, const double *St, const double *ds,
                           const double *dl, const double *dcoal, double *frag_volume,
                           const double *x_plus_y, double *rand, int64_t n, double vmin,
                           double nfmax, double *Rf, double *Rs, double *Rd, double tol,
                           const double consts[4]);
/* fragmentation_methods.py:305-319: Ec[i] = 1 where dl[i] < 0.4 mm */
int sdm_ll82_coalescence_check(sdm_ctx *ctx, double *Ec, const double *dl, int64_t n);

/* terminal_velocity_methods.py:32-66: Rogers & Yau's three regimes (consts = {SMALL_K, MEDIUM_K,
 * LARGE_K, SMALL_R_LIMIT, MEDIUM_R_LIMIT}) and the user-defined power series
 * values[i] = sum_j prefactors[j] * radius[i]^(3 powers[j]) (host arrays of num_terms <= 16)      */
int sdm_terminal_velocity(sdm_ctx *ctx, double *values, const double *radius, int64_t n,
                          const double consts[5]);
int sdm_power_series(sdm_ctx *ctx, double *values, const double *radius, int64_t n,
                     int num_terms, const double *prefactors, const double *powers);

/* ---- f-3 displacement, PySDM/backends/impl_numba/methods/displacement_methods.py ---------- */
/* :14-129: displacement[dim, :] from the Arakawa-C Courant field of direction `dim` (shape =
 * grid with one more point along dim; row-major) interpolated to the SD's position in its cell;
 * scheme 0 = ImplicitInSpace, 1 = ExplicitInSpace (physics/particle_advection/);
 * displacement, cell_origin, position_in_cell are (n_dims, n_sd) row-major                     */
int sdm_calculate_displacement(sdm_ctx *ctx, int dim, int n_dims, int scheme,
                               double *displacement, const double *courant,
                               const int64_t *courant_shape, const int64_t *cell_origin,
                               const double *position_in_cell, int64_t n_sd, double n_substeps);
/* :131-166,192-218: SDs moving down through `level` (in cells, along the last dimension) get
 * idx[i] = n_sd and healthy[0] = 0; *rainfall_mass (host) = sum |m| * n over them; syncs        */
int sdm_flag_precipitated(sdm_ctx *ctx, const int64_t *cell_origin,
                          const double *position_in_cell, const double *water_mass,
                          const int64_t *multiplicity, int64_t *idx, int64_t length,
                          int64_t n_sd, int n_dims, int64_t *healthy, double level,
                          const double *displacement, double *rainfall_mass);
/* :168-190,220-238: SDs below 0 or above `top` along the last dimension are flagged out         */
int sdm_flag_out_of_column(sdm_ctx *ctx, const int64_t *cell_origin,
                           const double *position_in_cell, int64_t *idx, int64_t length,
                           int64_t n_sd, int n_dims, int64_t *healthy, double top);
/* Storage.floor(other) into an int64 storage and float -= int64 (storage_impl.py:44-47 and
 * storage.py:76-78 as used by dynamics/displacement.py:146-150)                                */
int sdm_floor_to_i64(sdm_ctx *ctx, int64_t *out, const double *a, int64_t n);
int sdm_subtract_i64(sdm_ctx *ctx, double *out, const int64_t *b, int64_t n);

/* ---- f-1 moments, PySDM/backends/impl_numba/methods/moments_methods.py:14-99 ------------ */
int sdm_moments(sdm_ctx *ctx, double *moment_0, double *moments, const int64_t *multiplicity,
                const double *attr_data, const int64_t *cell_id, const int64_t *idx,
                int64_t length, const double *ranks, int64_t n_ranks, int64_t n_cell,
                double min_x, double max_x, const double *x_attr,
                const double *weighting_attribute, double weighting_rank,
                int skip_division_by_m0);
/* moments_methods.py:100-182: per (bin, cell) moments; x_bins has n_bins + 1 edges, the first
 * bin with x_bins[k] <= x < x_bins[k+1] takes the SD; moment_0 / moments are (n_bins, n_cell)   */
int sdm_spectrum_moments(sdm_ctx *ctx, double *moment_0, double *moments,
                         const int64_t *multiplicity, const double *attr_data,
                         const int64_t *cell_id, const int64_t *idx, int64_t length, double rank,
                         const double *x_bins, int64_t n_bins, int64_t n_cell,
                         const double *x_attr, const double *weighting_attribute,
                         double weighting_rank);

/* ---- a-20 Storage element-wise ops, PySDM/backends/impl_numba/storage_impl.py ----------- */
/* out[i] = a[i] (op) b[i]  or  a[i] (op) scalar when b == NULL.  out may alias a.            */
#define SDM_EW_ADD 0
#define SDM_EW_SUB 1
#define SDM_EW_MUL 2
#define SDM_EW_DIV 3
#define SDM_EW_POW 4          /* sign(a) * |a| ** scalar  (storage_impl.py:75-78) */
#define SDM_EW_DIV_IF_NOT_ZERO 5
#define SDM_EW_FLOOR 6
#define SDM_EW_EXP 7
#define SDM_EW_ABS 8
#define SDM_EW_FILL 9
#define SDM_EW_ADD_MUL 10     /* out = a + scalar * b  (add_with_multiplier :19-21) */
#define SDM_EW_MOD 11         /* Python-style % (row_modulo :36-41) */
int sdm_elementwise_f64(sdm_ctx *ctx, int op, double *out, const double *a, const double *b,
                        double scalar, int64_t n);
int sdm_elementwise_i64(sdm_ctx *ctx, int op, int64_t *out, const int64_t *a, const int64_t *b,
                        int64_t scalar, int64_t n);
/* the path's transcendental functions, evaluated element-wise: out[i] = fn(a[i] [, b[i]]).
 * The reference takes them from NumPy / libm (storage_impl.py:75-78 `power`, :48-49 `exp`,
 * fragmentation_methods.py:12-48 log/exp/erf, physics/trivia.py:95-108 sinh/asinh/atanh); here
 * they are ONE double-only implementation (csrc/sdm_math.h) compiled into this library and into
 * the CPU checker alike, so that both return the same bits and breakup runs stay
 * integer-identical however long they are.  b is read for SDM_MATH_POW only.                  */
#define SDM_MATH_EXP 0
#define SDM_MATH_LOG 1
#define SDM_MATH_POW 2
#define SDM_MATH_SINH 3
#define SDM_MATH_ASINH 4
#define SDM_MATH_ATANH 5
#define SDM_MATH_ERF 6
#define SDM_MATH_LOG1P 7
int sdm_math_eval(sdm_ctx *ctx, int fn, double *out, const double *a, const double *b,
                  int64_t n);
/* reductions (storage_impl.py:24-31): *result is a host pointer; syncs. kind: 0 min, 1 max */
int sdm_reduce_f64(sdm_ctx *ctx, int kind, const double *a, int64_t n, double *result);

/* ---- the fused per-time-step path (perf path) -------------------------------------------
 * One call = one `Collision.__call__` (PySDM/dynamics/collisions/collision.py:174-234): all
 * sub-steps of one time step: counting sort if unsorted, on-the-fly PCG64, shuffle, pairing,
 * kernel/probability, (Ec/Eb/fragmentation,) adaptive dt, gamma, multiplicity/attribute update,
 * counters, compaction of zero-multiplicity super-droplets.                                  */
#define SDM_KERNEL_GOLOVIN 0    /* collision_kernels/golovin.py:14-16 ; kernel_param[0] = b */
#define SDM_KERNEL_GEOMETRIC 1  /* collision_kernels/geometric.py:15-22 ; [0] = PI * E_coll */
#define SDM_KERNEL_CONSTANT 2   /* collision_kernels/constantK.py ; [0] = a */
#define SDM_KERNEL_PARAMETERIZED 3 /* collision_kernels/impl/parameterized.py:8-30 (Electric,
                                      Hydrodynamic) ; kernel_berry_params, kernel_berry_unit */
#define SDM_KERNEL_SIMPLE_GEOMETRIC 4 /* collision_kernels/simple_geometric.py ; [0] = C */
#define SDM_KERNEL_LINEAR 5     /* collision_kernels/linear.py ; a + b (v_j + v_k), [0] = a [1] = b */
#define SDM_EC_CONST 0          /* coalescence_efficiencies/constEc.py ; ec_param[0] = Ec */
#define SDM_EC_BERRY1967 1      /* coalescence_efficiencies/berry1967.py */
#define SDM_EC_STRAUB2010 2     /* coalescence_efficiencies/straub2010.py:27-50 */
#define SDM_FRAG_ALWAYS_N 0     /* breakup_fragmentations/always_n.py ; frag_param[0] = n */
#define SDM_FRAG_EXPONENTIAL 1  /* breakup_fragmentations/exponential.py ; [0] = scale */
#define SDM_FRAG_STRAUB2010 2   /* breakup_fragmentations/straub2010.py */
#define SDM_FRAG_GAUSSIAN 3     /* gaussian.py ; frag_param = {mu, sigma} */
#define SDM_FRAG_FEINGOLD1988 4 /* feingold1988.py ; frag_param = {scale, fragtol} */
#define SDM_FRAG_SLAMS 5        /* slams.py */
#define SDM_FRAG_CONSTANT_MASS 6 /* constant_mass.py ; frag_param[0] = c */
#define SDM_FRAG_LOWLIST1982 7  /* lowlist82.py (straub_consts supplies CM, PI, Vedder's A, b) */
#define SDM_EC_LOWLIST1982 3    /* coalescence_efficiencies/lowlist1982.py */

typedef struct sdm_step_cfg {
  int64_t n_sd, n_cell, n_attr;
  double dt, dv;
  double dt_min, dt_max;      /* dt_coal_range after clamping (collision.py:115-116) */
  int32_t adaptive;
  int32_t substeps;           /* non-adaptive only */
  int32_t croupier_local;     /* 1 = local (per-cell), 0 = global + re-sort */
  int32_t optimized_random;   /* random_generator_optimizer.py:37-48 */
  int32_t enable_breakup;
  int32_t handle_all_breakups;
  int32_t kernel, ec, frag;   /* SDM_KERNEL_*, SDM_EC_*, SDM_FRAG_* */
  int32_t mass_attr;          /* row of `attributes` holding "signed water mass" */
  double kernel_param[2];
  double ec_param[2];
  double eb_const;            /* breakup_efficiencies/constEb.py */
  double frag_param[2];
  double frag_vmin, frag_nfmax; /* nfmax < 0 == None */
  double rho_w, sgm_w;
  double straub_consts[6];    /* as sdm_straub_fragmentation */
  double berry_params[13];
  double berry_unit;
  double kernel_berry_params[13]; /* SDM_KERNEL_PARAMETERIZED: the 13 parameters of its efficiency */
  double kernel_berry_unit;
  int64_t max_multiplicity;
  uint64_t rng_state_inc[4];  /* PCG64 state/inc of seed (numpy.random.PCG64(seed).state) */
  int64_t gk_table_len;       /* Gunn-Kinzer table length (0 if unused) */
  double gk_factor;
} sdm_step_cfg;

/* ---- sharding of a multi-cell domain over several processes (one per GPU) -------------------
 * Pairs never span cells (pair_methods.py:34-55), so cells can be divided among processes; what
 * remains global in the reference's algorithm is (i) the position of a cell's super-droplets in
 * the sorted permutation - it indexes the random stream (random_generator_optimizer.py:37-48,
 * index_methods.py:32-43) and, through `normalize`'s raw look-up (collisions_methods.py:633-662),
 * the probabilities -, (ii) the order of the cells (`cell_idx.sort_by_key(dt_left)`,
 * collision.py:183) and the working length (`adaptive_sdm_end`, collisions_methods.py:313-328),
 * (iii) the compaction that fills the holes left by dead super-droplets with super-droplets from
 * the end of the permutation (collisions_methods.py:664-680).  In sharded mode every process holds
 * arrays of the GLOBAL shape (ids, positions, cell_start, cell_idx are global; 288 GB of HBM make
 * that free) but computes only the cells it owns, and of its permutation only the segments of
 * those cells are exact: every other segment holds as many ids as the cell has members, each an
 * id of that cell - all the replicated compaction and the stable counting sort need to put the
 * owned segments into the reference's order.  The library calls `exchange` where global data is
 * needed.  After every sub-step ONE all-reduce: on the per-cell adaptive route (local croupier,
 * cells of at most 6144) a MINIMUM over 2 n_cell doubles - per cell id the minimum of the optimal
 * sub-step over the cell's pairs (+inf from everybody but the owner), per segment of the
 * permutation minus the number of super-droplets that died in it - from which every process
 * derives dt_left, the order of the cells and the working length by itself; on the other routes a
 * SUM over n_cell + 1 + shard_world doubles (masked dt_left; how many died, in total and per
 * process).  And - only when one died - a sum of rank-disjoint slices holding the POSITIONS of the
 * dead (exactly as many int64 as died; ordered by segment on the per-cell route, by process on the
 * others): every process flags those positions and runs the compaction and the counting sort on
 * its own copy.  No super-droplet payload and no permutation crosses processes in a collision step.
 * The concatenation of the owned cells equals the one-process result bit for bit.            */
#define SDM_XCHG_SUM_F64 1 /* buffer = device double[count]: in-place sum over all processes */
#define SDM_XCHG_SUM_I64 2 /* buffer = device int64[count] */
#define SDM_XCHG_MIN_F64 3 /* buffer = device double[count]: in-place minimum over all processes */
/* must order the collective after the work already enqueued on the ctx stream and make its result
 * visible to work enqueued later on that stream; returns 0 on success */
typedef int (*sdm_exchange_fn)(void *user, int what, void *device_buffer, int64_t count);

/* The same exchanges issued by the library itself: RCCL all-reduces (xGMI between the GPUs of a
 * node) on the context's stream, no host code inside the sub-step loop.  The host keeps the
 * bootstrap: ONE process calls sdm_comm_unique_id, the id reaches the others by whatever the host
 * has (torch.distributed's store, MPI, a file), EVERY process then calls sdm_comm_init (collective).
 * While a context has a communicator, `exchange` of sdm_step_state / sdm_disp_shard is not called
 * and may be NULL.  sdm_shard_set_comm adopts a communicator (ncclComm_t) the caller created; NULL
 * returns to the callback.  RCCL is looked up at run time (the copy the process has loaded, else
 * librccl.so.1): the library neither links against it nor needs it otherwise.
 * (The reference has no counterpart: PySDM is a one-process code.)                              */
#define SDM_COMM_ID_BYTES 128
int sdm_comm_unique_id(uint8_t *id);  /* [SDM_COMM_ID_BYTES], host */
int sdm_comm_init(sdm_ctx *ctx, const uint8_t *id, int rank, int world);
int sdm_shard_set_comm(sdm_ctx *ctx, void *rccl_comm);
int sdm_comm_destroy(sdm_ctx *ctx);

typedef struct sdm_step_state {
  int64_t *idx;               /* [n_sd] current permutation */
  int64_t *tmp_idx;           /* [n_sd] counting-sort / shuffle double buffer */
  int64_t *multiplicity;      /* [n_sd] */
  double *attributes;         /* [n_attr, n_sd] extensive attributes */
  int64_t *cell_id;           /* [n_sd] */
  int64_t *cell_idx;          /* [n_cell] */
  int64_t *cell_start;        /* [n_cell + 1] */
  double *dt_left;            /* [n_cell] */
  double *stats_dt_min;       /* [n_cell] */
  int64_t *stats_n_substep;   /* [n_cell] */
  int64_t *collision_rate, *collision_rate_deficit, *coalescence_rate;  /* [n_cell] */
  int64_t *breakup_rate, *breakup_rate_deficit;                        /* [n_cell] or NULL */
  const double *gk_a, *gk_b;  /* Gunn-Kinzer table (or NULL) */
  /* device control block, int64[8]: {valid_n_sd, working_length, sorted, healthy, n_overflow,
   * candidate pairs processed so far (single-cell non-adaptive steps), largest cell, events};
   * kept device-resident between calls.  Word 7: low byte = device-side error code (0 = none;
   * 1 = a cell larger than the per-cell kernel's capacity, 2 = the compaction kernel's grid
   * barrier timed out; bit value 4 is used inside the library: `sorted` was claimed but cell_start
   * does not span exactly the live super-droplets - the call returns SDM_E_ARG before any
   * collision is computed),
   * bit 8 = "a cell's stats_dt_min became equal to dt_min" - the data-level event behind the
   * reference's warning "adaptive time-step reached dt_min" (collision.py:276-277): the caller
   * evaluates `amin(stats_dt_min) == dt_min`, warns, and clears the bit; n_overflow likewise
   * stands for the "overflow" warning (collisions_methods.py:196-199) */
  int64_t *ctl;
  /* optional mirror kept by the library, caller-owned like all state (NULL = do not use):
   * [n_sd] x 32 B.  Records are {int64 multiplicity, double mass} (16-B stride) or, with the
   * geometric kernel / Berry / Straub breakup, {multiplicity, mass, radius, terminal velocity}
   * (32-B stride): one random line per gather, derived attributes evaluated once per change.
   * (Re)initialised from the SoA columns when SDM_STEP_FRESH_CTL is passed.                   */
  void *nm;
  /* library's bookkeeping between calls: live super-droplets as of the last read-back (also the
   * one every adaptive sub-step ends with), so that a single-cell adaptive step need not ask the
   * device again (-1 = unknown; set it to -1 whenever SDM_STEP_FRESH_CTL is passed)           */
  int64_t known_valid;
  uint64_t rng_offset;        /* doubles already drawn from the coll. stream (host-tracked) */
  uint64_t rng_offset_breakup;/* doubles already drawn from the proc/frag streams */
  /* sharded mode (see above; all NULL = this process owns every cell).  Any croupier, any cell
   * size: cells of at most 6144 super-droplets under the local croupier take the per-cell
   * kernels, everything else the generic ones, as in a one-process run. */
  const uint8_t *cell_owned;  /* [n_cell] by cell id: 1 = computed by this process */
  sdm_exchange_fn exchange;
  void *exchange_user;
  double *xchg_cells;         /* [max(4 n_cell, n_cell + 1 + shard_world)] scratch for the per-cell
                                 exchanges (adaptive steps of the per-cell kernels: two buffers of
                                 {cell minima, minus the deaths per segment}, reduced with MIN) */
  int64_t *xchg_idx;          /* [n_sd] scratch for the exchange of dead positions */
  int32_t shard_rank, shard_world; /* this process's place among the processes (slices above) */
  /* sharded run with a sharded displacement step (sdm_disp_shard below): `cell_id` then holds, for
   * ids that stand in other processes' positions, the cell of the super-droplet truly there; the
   * one place the reference reads a cell id by RAW id - `normalize` indexes cell_id with the pair
   * number (collisions_methods.py:430-442) - needs every id's own cell: this column.  NULL = cell_id */
  const int64_t *cell_id_by_id;
} sdm_step_state;

typedef struct sdm_step_result {
  int64_t n_substeps;         /* sub-steps executed in this call */
  int64_t n_pairs;            /* candidate pairs processed (sum of working_length // 2) */
  int64_t valid_n_sd;         /* live super-droplets after the call (-1 if not read back) */
  int64_t idx_swapped;        /* 1 if state->idx / state->tmp_idx exchanged roles */
  uint64_t rng_offset, rng_offset_breakup; /* updated stream positions */
  int64_t ctl[8];             /* the control block after the call (valid with SDM_STEP_READ_BACK) */
} sdm_step_result;

/* flags: bit 0 = read the control block back (fills result->valid_n_sd; synchronises);
 *        bit 1 = state->ctl was freshly written by the host (first call / after host-side edits);
 *        bit 2 = with bit 1: multiplicities and attributes are untouched since the last fused call
 *                (only the permutation / cell ids changed, e.g. by a displacement), so the mirror
 *                state->nm is still current and is not rebuilt                                 */
/* Adaptive steps: every sub-step takes at least dt_min off the dt_left of each cell still in the
 * working range (collisions_methods.py:355-374), so a time step has at most ceil(dt / dt_min) + 1
 * sub-steps (+ 1 for the rounding of the subtractions).  The reference's loop has no bound
 * (collision.py:182 `while working_length != 0`) and never ends on a state whose cell_start does
 * not belong to its permutation; here more sub-steps than that return SDM_E_STATE with the
 * control block in sdm_last_error().                                                           */
#define SDM_STEP_READ_BACK 1
#define SDM_STEP_FRESH_CTL 2
#define SDM_STEP_MIRROR_VALID 4
int sdm_collision_step(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *state,
                       sdm_step_result *result, int flags);
/* n_steps time steps back to back (`Particulator.run(n_steps)` with the collision dynamic alone,
 * PySDM/particulator.py:50-56); state->idx / tmp_idx are exchanged in place as needed, result
 * holds totals (idx_swapped = parity over the run).                                          */
int sdm_collision_run(sdm_ctx *ctx, const sdm_step_cfg *cfg, sdm_step_state *state,
                      sdm_step_result *result, int flags, int64_t n_steps);

/* ---- the fused displacement step (f-3) ------------------------------------------------------
 * One call = one `Displacement.__call__` (PySDM/dynamics/displacement.py:100-121): all its
 * sub-steps, each { displacement of every dimension, sedimentation, position update,
 * precipitation flagging + removal, out-of-column flagging + removal, cell origin / position
 * renormalisation, periodic boundary, cell id } in four launches plus the two gated compactions,
 * without returning to the host in between.                                                   */
typedef struct sdm_disp_cfg {
  int64_t n_sd;
  int32_t n_dims;               /* 1..3 */
  int32_t scheme;               /* 0 ImplicitInSpace, 1 ExplicitInSpace */
  int32_t enable_sedimentation;
  int32_t n_substeps;
  int64_t grid[3];              /* cells per dimension (unused dimensions: 1) */
  int64_t strides[3];           /* mesh.strides: cell id = sum origin[d] * strides[d] */
  double dt_over_dz;            /* dt / n_substeps / dz (displacement.py:131-133) */
  double level;                 /* precipitation_counting_level_index */
} sdm_disp_cfg;

typedef struct sdm_disp_state {
  const double *courant[3];     /* component d: grid shape with grid[d] + 1 points along d */
  double *displacement;         /* (n_dims, n_sd) */
  double *position_in_cell;     /* (n_dims, n_sd) */
  int64_t *cell_origin;         /* (n_dims, n_sd) */
  int64_t *cell_id;             /* (n_sd) */
  const double *fall_velocity;  /* (n_sd) "relative fall velocity"; NULL without sedimentation */
  const double *water_mass;     /* (n_sd) */
  const int64_t *multiplicity;  /* (n_sd) */
  int64_t *idx;                 /* (n_sd) permutation; removed super-droplets are compacted out */
  int64_t *ctl;                 /* 8 device words {valid n_sd, working length, sorted, healthy,..}
                                   set by the caller: {n_valid, n_valid, *, 1}; on return valid =
                                   work = survivors, sorted = 0, healthy = 1 */
} sdm_disp_state;

/* *rainfall_mass = mass of water that left through the counting level in this step (sum over the
 * sub-steps, in sub-step order), *valid_n_sd = surviving super-droplets; synchronises once.      */
int sdm_displacement_step(sdm_ctx *ctx, const sdm_disp_cfg *cfg, const sdm_disp_state *state,
                          double *rainfall_mass, int64_t *valid_n_sd);

/* ---- the displacement step of a sharded run ------------------------------------------------------
 * Between two sharded collision steps (see "sharding" above) a process holds exact data only for
 * the super-droplets of its own cells: their rows in the columns, and their POSITIONS in the
 * permutation.  Every other position holds a placeholder - some id that is not one of its own,
 * whose cell_id entry is the cell of the super-droplet that is truly there (all the replicated
 * compaction and counting sort read of it).  This entry point keeps exactly that:
 *   - a super-droplet is moved (all n_substeps of it: displacement depends on nothing but the
 *     droplet and the replicated Courant field) by the process that owns its cell when the call
 *     begins; rows of other super-droplets are not touched;
 *   - removal (precipitation, out of the column): the owners' dead POSITIONS (for precipitation
 *     also the masses, as bit patterns) are summed as rank-disjoint slices, as in the collision
 *     step, and every process runs the reference's compaction on its own permutation.  Positions
 *     are the global names: a filler taken from the tail lands in the same hole on every process;
 *   - at the end, per super-droplet whose cell changed: {position and id (one word), new cell} to
 *     everybody (cell_id_by_id[id] takes the new cell, and so does the cell_id entry of the
 *     placeholder at that position), and for those that changed OWNER the row itself -
 *     {position, id, new cell, multiplicity, cell origin; attributes and position in cell as bit
 *     patterns} - which the new owner stores under the true id, at the true position (the
 *     placeholders involved trade places among themselves).  One sum of exactly the int64 words
 *     listed; nothing the size of a column;
 *   - how many words each list has: every sub-step begins with ONE sum of counts (2 * world
 *     doubles: how many each process will list as precipitated / as out of the column - both are
 *     decided by where the move has put the droplets; 4 * world in the last sub-step, which adds
 *     the two lists of the previous item: the cells are known by then as well).  That is the one
 *     host round trip of a sub-step;
 *   - removed super-droplets keep moving in the reference (the displacement kernels run over the
 *     raw columns, and `normalize` reads cell ids by raw id, dead or alive): the process that
 *     owned one when it was removed keeps moving it, wherever it goes, and announces its cells
 *     like the others' (position -1 in the list) - if its id is below (n_sd + 1) / 2: pair
 *     numbers end there, nobody ever asks for the cell of a removed id beyond.  `role` records
 *     who is whose.
 * Everything - ids, multiplicities, attributes, positions, cells of the owned super-droplets; the
 * permutation after the next sharded collision step; the rainfall (the masses of the precipitated
 * travel with their positions, and every process adds them up in the one-process order) - is the
 * one-process result bit for bit.
 * The reference has no counterpart (PySDM is a one-process code); the step computed is
 * PySDM/dynamics/displacement.py:100-153 as above. */
typedef struct sdm_disp_shard {
  const uint8_t *cell_owned;  /* [n_cell] by cell id: 1 = this process's cell */
  int64_t n_cell;
  sdm_exchange_fn exchange;
  void *exchange_user;
  int32_t shard_rank, shard_world;
  double *xchg_counts;        /* [4 * shard_world] scratch for the counts */
  int64_t *xchg_words;        /* [word_capacity] scratch for positions / rows */
  int64_t word_capacity;      /* SDM_E_ARG if a step needs more (n_sd * (6 + 2 * n_dims + n_attr)
                                 always suffices) */
  int64_t *cell_id_by_id;     /* [n_sd] every id's own cell (see sdm_step_state; kept up to date
                                 for the alive and for ids below (n_sd + 1) / 2); initially a
                                 copy of cell_id */
  /* [n_sd] kept by the caller between calls, written by the library: 0 = not this process's,
   * 1 = alive and in one of its cells, 2 = removed while it was (kept moving here).  role_ready = 0:
   * the library fills it from cell_id_by_id and the permutation first (and sets role_ready) */
  uint8_t *role;
  int32_t role_ready;
  int64_t *multiplicity;      /* [n_sd] writable: rows of arriving super-droplets */
  double *attributes;         /* [n_attr, n_sd] */
  int32_t n_attr;
  /* out, this call: super-droplets of this process whose cell changed; of those, how many changed
   * owner; rows received; int64 words handed to `exchange`; super-droplets removed (all ranks) */
  int64_t n_moved, n_left, n_arrived, n_words, n_removed;
} sdm_disp_shard;

int sdm_displacement_step_sharded(sdm_ctx *ctx, const sdm_disp_cfg *cfg,
                                  const sdm_disp_state *state, sdm_disp_shard *shard,
                                  double *rainfall_mass, int64_t *valid_n_sd);

#ifdef __cplusplus
}
#endif
#endif /* SDM_HIP_H */
