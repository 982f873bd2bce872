/*
 * mmsbm_hip.h -- C ABI of the MI355X-native (gfx950) EM core for the Mixed-Membership
 * Stochastic Block Model.
 *
 * This is the drop-in boundary for the hot path of eudald-seeslab/mmsbm (paths below
 * are relative to the reference checkout):
 *
 *   src/backend.py:16-22              load_backend(name) -> (compute_omegas,
 *                                     update_coefficients, prod_dist, name)
 *   src/kernels_numpy.py:21-36        compute_omegas(data, theta, eta, pr)
 *   src/kernels_numpy.py:43-79        update_coefficients(data, theta, eta, pr)
 *   src/kernels_numpy.py:86-96        prod_dist(data, theta, eta, pr)
 *   src/expectation_maximization.py:118-120,152-155,157-167
 *                                     normalize_with_d / normalize_with_self / compute_likelihood
 *   src/mmsbm.py:243-256              the per-restart EM loop
 *
 * The reference has no native layer; a Python module `kernels_hip` binds these entry
 * points with ctypes.  Conventions:
 *
 *   - plain C types only; every host buffer is owned by the caller, C-contiguous,
 *     float64 / int32, and is never retained after the call returns;
 *   - host-side parameter layouts are the reference's: theta (U,K), eta (I,L),
 *     pr (K,L,R), row-major;
 *   - every function returns 0 on success or an MMSBM_E_* code; the message for the
 *     calling thread's last failure is mmsbm_hip_last_error();
 *   - a context is bound to one device and one stream and must be driven by one host
 *     thread at a time; distinct contexts (e.g. one per GPU) are independent;
 *   - there is no CPU fallback: without a usable HIP device every call fails.
 */
#ifndef MMSBM_HIP_H
#define MMSBM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMSBM_HIP_ABI_VERSION 1

enum {
  MMSBM_OK = 0,
  MMSBM_E_INVALID = 1,     /* bad argument (null pointer, index out of range, size) */
  MMSBM_E_HIP = 2,         /* a HIP runtime call failed                              */
  MMSBM_E_NODEVICE = 3,    /* no usable gfx950 device                                */
  MMSBM_E_UNSUPPORTED = 4, /* shape outside what the kernels are instantiated for    */
  MMSBM_E_TOOLARGE = 5,    /* output would exceed the caller's capacity / byte cap   */
  MMSBM_E_INTERNAL = 6
};

typedef struct mmsbm_hip_ctx mmsbm_hip_ctx;

/* ---- library / device ---------------------------------------------------------- */
int mmsbm_hip_abi_version(void);
/* Identity of the kernel sources this binary was compiled from: the first 16 hex digits of the SHA-256 over
 * every file under mmsbm_amd/csrc (name, NUL, contents; in name order) -- what mmsbm_amd/build.py computes
 * and bench.py / profiles/pmc_summary.json record, so that a measurement can say which kernels really ran.
 * "unknown" when the library was built without the build script. */
const char *mmsbm_hip_build_id(void);
const char *mmsbm_hip_last_error(void);
int mmsbm_hip_device_count(int *count);
/* name: caller buffer of name_len bytes; arch e.g. "gfx950:sramecc+:xnack-". */
int mmsbm_hip_device_info(int device, char *name, int name_len, int *compute_units,
                          int64_t *global_mem_bytes);
/* PCI bus id of the device ("0000:05:00.0", hipDeviceGetPCIBusId): with the name, what tells the ranks of a
 * multi-GPU job apart -- bench.py prints it per rank so that a line proves which GPUs it ran on. */
int mmsbm_hip_device_pci(int device, char *bus_id, int bus_id_len);
/* Device memory that is free right now / in total (hipMemGetInfo): what batches of restart
 * slots are sized from -- other contexts and processes on the same GPU already count. */
int mmsbm_hip_device_mem(int device, int64_t *free_bytes, int64_t *total_bytes);

/* ---- context = (device, encoded training triples) -------------------------------- */
/* Replaces the per-call re-gathering of `data` in src/kernels_numpy.py:26-28 and the
 * degree pre-computation of src/mmsbm.py:100-111.  user/item/rating: n_obs int32 ids in
 * [0,U) / [0,I) / [0,R).  Uploads, sorts (rating,item)-major and user-major, builds the
 * CSR offsets and degrees once.  swap_sides != 0 lets the library pair ratings with the
 * users instead of the items when that is the smaller table (results are identical up
 * to summation order); pass 0 for the default, -1 for "choose automatically". */
int mmsbm_hip_create(int device, int64_t n_obs, int32_t n_users, int32_t n_items,
                     int32_t n_ratings, int32_t k_groups, int32_t l_groups,
                     const int32_t *user, const int32_t *item, const int32_t *rating,
                     int swap_sides, mmsbm_hip_ctx **out);
int mmsbm_hip_destroy(mmsbm_hip_ctx *ctx);
/* dims[0..7] = n_obs, U, I, R, K, L, n_pairs (distinct (item,rating) pairs), swapped */
int mmsbm_hip_dims(const mmsbm_hip_ctx *ctx, int64_t dims[8]);
/* d_u (U) and d_i (I): rows per user / item, floored at 1 (src/mmsbm.py:106-111). */
int mmsbm_hip_degrees(const mmsbm_hip_ctx *ctx, int64_t *d_user, int64_t *d_item);

/* ---- parameters (device resident between calls) ---------------------------------- */
int mmsbm_hip_set_params(mmsbm_hip_ctx *ctx, const double *theta, const double *eta,
                         const double *pr);
int mmsbm_hip_get_params(mmsbm_hip_ctx *ctx, double *theta, double *eta, double *pr);
/* The random initialisation of src/mmsbm.py:224-233 ON the device, bit for bit:
 * theta0 = rng.random((U,K)) / d_u, then eta0 = rng.random((I,L)) / d_i, drawn from the PCG64
 * stream (numpy's default_rng) whose state is pcg64_state = { state_hi, state_lo, inc_hi, inc_lo }
 * (numpy: bit_generator.state["state"]["state" | "inc"], split in 64-bit halves).  pr is the
 * third draw -- p0 (K,L,R), already normalised -- which the caller makes on the host from the
 * same stream advanced by U*K + I*L draws (it is tiny).  Equivalent to
 * set_params(theta0, eta0, pr) without generating or moving the two big tables on the host. */
int mmsbm_hip_init_params(mmsbm_hip_ctx *ctx, const uint64_t pcg64_state[4], const double *pr);
/* Host-only: n doubles of that stream starting `offset` draws in (what Generator.random gives
 * after advancing); lets the generator be checked against numpy without a GPU. */
int mmsbm_hip_pcg64_doubles(const uint64_t pcg64_state[4], uint64_t offset, int64_t n, double *out);

/* ---- restart slots: several restarts of one training set in one context ------------- */
/* The reference runs its `sampling` restarts as independent processes over the same triples
 * (src/mmsbm.py:182-185; batching them is the TODO of README.md:188).  A context can hold
 * n_slots independent parameter sets ("slots") that share the sorted triples on the device:
 * mmsbm_hip_em_iterate advances ALL slots with one set of kernel launches (in the two triple
 * passes one group of lanes walks a segment for all slots: one index stream, one contiguous row
 * gather for every restart), every other entry point (set/get_params, update_coefficients, likelihood,
 * compute_omegas, prod_dist) acts on the SELECTED slot.  Slots never interact: slot s holds
 * exactly what a one-slot context given the same parameters would hold, bit for bit.
 * set_slots drops all parameters (set_params must follow for every slot) and selects slot 0;
 * a new context has one slot. */
int mmsbm_hip_set_slots(mmsbm_hip_ctx *ctx, int n_slots);
int mmsbm_hip_select_slot(mmsbm_hip_ctx *ctx, int slot);
/* Any output may be NULL.  bytes_per_slot: device memory one more slot costs (a failed
 * set_slots -- out of device memory -- leaves the context with ONE slot and no parameters). */
int mmsbm_hip_slots(const mmsbm_hip_ctx *ctx, int *n_slots, int *selected,
                    int64_t *bytes_per_slot);

/* ---- the hot loop: src/mmsbm.py:243-250 ------------------------------------------- */
/* n_iters x { update_coefficients; theta = n_theta/d_u; eta = n_eta/d_i;
 * pr = normalize_with_self(n_pr) } entirely on the device; enqueues on the context's
 * stream and returns without synchronising. */
int mmsbm_hip_em_iterate(mmsbm_hip_ctx *ctx, int n_iters);
int mmsbm_hip_synchronize(mmsbm_hip_ctx *ctx);

/* One un-normalised M-step from the current parameters (src/kernels_numpy.py:43-79).
 * Outputs in host layout; the context's parameters are left unchanged.  Any output
 * pointer may be NULL. */
int mmsbm_hip_update_coefficients(mmsbm_hip_ctx *ctx, double *n_theta, double *n_eta,
                                  double *n_pr);

/* src/expectation_maximization.py:157-167 on the current parameters. */
int mmsbm_hip_likelihood(mmsbm_hip_ctx *ctx, double *out);
/* What a finished restart hands back (src/mmsbm.py:256-269): the likelihood and the parameters of the
 * selected slot in one call -- the same values as mmsbm_hip_likelihood + mmsbm_hip_get_params, with the
 * parameter download and its host-side unpacking overlapped with the likelihood kernels (at K = L = 50 and
 * 10M ratings each of the two takes ~33 ms).  theta / eta / pr may be NULL. */
int mmsbm_hip_result(mmsbm_hip_ctx *ctx, double *theta, double *eta, double *pr, double *likelihood);

/* src/kernels_numpy.py:21-36: (N,K,L) tensor in the ORIGINAL row order of the triples
 * given to create().  Contract / test use only: refuses if N*K*L > capacity_elems. */
int mmsbm_hip_compute_omegas(mmsbm_hip_ctx *ctx, double *out, int64_t capacity_elems);

/* src/kernels_numpy.py:86-96 for n_pairs (user,item) pairs; out is (n_pairs, R). */
int mmsbm_hip_prod_dist(mmsbm_hip_ctx *ctx, int64_t n_pairs, const int32_t *user,
                        const int32_t *item, double *out);

/* ---- predict / score on the device: src/mmsbm.py:297-315 and 488-539 ------------------ */
/* A session over n_rows test triples (ids as in create(); rating = the true rating index;
 * rating_weights: R doubles, the values the reference multiplies the distribution with --
 * its `self.ratings`, src/mmsbm.py:95,518):
 *   begin  uploads the rows;
 *   add    evaluates prod_dist for the SELECTED slot's current parameters, adds it to the
 *          running sum over restarts (in call order, which is numpy's order for
 *          np.array(rats).mean(axis=0)) and returns that restart's indicators;
 *   finish divides by the number of adds, returns the mean distribution (n_rows x R, may be
 *          NULL) and ITS indicators, and closes the session.
 * stats[6] = { rows kept (distribution not all zero), argmax == real, |argmax - real| <= 1,
 *              sum |argmax - real|, real == round(P . w), sum |P . w - real| }, from which
 * accuracy = [1]/[0], one_off = [2]/[0], mae = 1 - [4]/[0], s2 = [3], s2pond = [5]
 * (src/mmsbm.py:530-539).  Counts are exact; argmax takes the first maximum like np.argmax. */
int mmsbm_hip_predict_begin(mmsbm_hip_ctx *ctx, int64_t n_rows, const int32_t *user,
                            const int32_t *item, const int32_t *rating,
                            const double *rating_weights);
int mmsbm_hip_predict_add(mmsbm_hip_ctx *ctx, double stats[6]);
int mmsbm_hip_predict_finish(mmsbm_hip_ctx *ctx, double *mean_dist, double stats[6]);

/* ---- measurement ------------------------------------------------------------------ */
/* Runs n_iters EM iterations bracketed by HIP events on the context's stream; returns
 * the elapsed device time of the whole region in milliseconds (synchronises). */
int mmsbm_hip_time_iterations(mmsbm_hip_ctx *ctx, int n_iters, float *elapsed_ms);
/* Number of kernel launches in one EM iteration and their names. */
int mmsbm_hip_kernel_count(void);
const char *mmsbm_hip_kernel_name(int index);
/* Runs n_iters iterations with a HIP event pair around EVERY kernel launch (on the
 * context's stream) and returns the mean duration per launch, in microseconds, for each
 * of the mmsbm_hip_kernel_count() kernels, plus launches per iteration. */
int mmsbm_hip_profile_iterations(mmsbm_hip_ctx *ctx, int n_iters, float *mean_us,
                                 int *launches_per_iter);
/* Algorithmic bytes (read, written) of kernel `index` for this context's shapes; the
 * accounting is stated in DESIGN.md. */
int mmsbm_hip_kernel_bytes(const mmsbm_hip_ctx *ctx, int index, int64_t *bytes_read,
                           int64_t *bytes_written);
/* Tuning aid: launches ONE stage of the iteration (index as in mmsbm_hip_kernel_name) `reps`
 * times back to back on the context's stream and returns the mean time per launch.  The
 * stage's outputs overwrite scratch/next buffers: call set_params again before trusting the
 * context's state. */
int mmsbm_hip_time_stage(mmsbm_hip_ctx *ctx, int stage, int reps, float *mean_us);
/* The library chooses its kernels from the shape and the data; these few switches exist because the tests compare
 * forms with each other (none is needed for normal use, and nothing that was measured and rejected is kept behind a
 * switch -- EXPERIMENTS.md has those): "graph" 0/1 (the same switch as mmsbm_hip_set_graph_mode), "fused" 0/1 (two
 * launches per iteration instead of four: small problems; 1 is refused where the form does not exist), "mfma" 0/1/2
 * (the pair stage on the matrix cores, v_mfma_f64_16x16x4_f64: 0 the vector-ALU form, 1 on -- the one-block kernel for
 * K, L <= 64, else the blocked kernels --, 2 the blocked kernels whatever the shape; create() turns it on for
 * K x L > 1024; environment MMSBM_HIP_NO_MFMA=1 keeps it off), "quad" 0/1 (vector-ALU form of long rows: the A launch
 * as a persistent four-unit pipeline), "lik_fast" 0/1/2 (likelihood: a logarithm per element / logarithm tables, a
 * group of lanes per triple / 2 = the default: a wave per pair or a lane per triple where those apply), "lik_g"
 * 0/1/2/4/8 (lanes per triple of the table form, 0 = automatic), "predict_fast" 0/1 (prod_dist / predict through the
 * table of p_r eta_i over every (item, rating) combination -- the default where the rows are not far fewer than the
 * items -- or through the one-thread-per-row kernels), "nt_out" 0..15 (bits: non-temporal stores of the T / A rows,
 * of the theta' rows, non-temporal loads of the segments' own rows; 8: whatever the data; results are bitwise the same),
 * "a_units" 0..16 (matrix-core pair stage only: 64-pair units per workgroup of the A launch, which writes rows only and
 * so walks the units in runs of its own length; 0 = the library's choice -- the length that fills its last round of
 * workgroups; the rows are bitwise the same for every value). */
int mmsbm_hip_set_option(mmsbm_hip_ctx *ctx, const char *name, double value);
/* Reads a switch back; also the read-only "launches" (2 or 4: what the next iteration takes), "ranges_pairs" / "ranges_users" (ranges the XCD-local work
 * list of that pass uses, 1 = off), "items_pairs" / "items_users" (work items, 0 = segments as
 * they are), "splits_pairs" / "splits_users" (segments cut into pieces), "fused_split" (bit 0 / 1: whole-segment lists of
 * the two-launch form built for the pair / user side), "chunk_pairs" (pairs per pair-stage workgroup at most), "n_chunks"
 * (pair-stage workgroups = slabs, padding included), "a_chunks" (workgroups of the matrix-core A launch when it walks
 * runs of its own; 0: the T + S launch's list serves) and "wide" (1: K, L beyond the 64-pair LDS stage -- the vector form of the pair stage is then the plain
 * wide-row kernels; they run when "mfma" reads 0, the blocked matrix-core kernels when it reads 2). */
int mmsbm_hip_get_option(const mmsbm_hip_ctx *ctx, const char *name, double *value);
/* How em_iterate launches: 0 (default) = eager launches on the context's stream; 1 = replay
 * a captured hipGraph of two iterations. */
int mmsbm_hip_set_graph_mode(mmsbm_hip_ctx *ctx, int enabled);

/* ---- host-only helpers (no device needed) ------------------------------------------- */
/* The sorted CSR-style layout create() uploads, exposed so it can be checked on a
 * machine without a GPU.  which: 0 pair_off, 1 pair_user, 2 pair_item, 3 rating_off,
 * 4 user_off, 5 user_pair, 6 item_off, 7 item_pairs, 8 item_deg, 9 chunk_off,
 * and arrays of 4-int records: 10 chunks and 11 mv_chunks (rating, q_begin, q_end, 0),
 * 12 / 13 work items of the pair / user segments (segment, begin, end, partial slot or -1;
 * empty when no segment is longer than 64 triples), 14 / 15 split segments (segment, first
 * partial slot, pieces, 0).  Pass out == NULL to query count. */
typedef struct mmsbm_hip_layout mmsbm_hip_layout;
int mmsbm_hip_layout_build(int64_t n_obs, int32_t n_users, int32_t n_items,
                           int32_t n_ratings, const int32_t *user, const int32_t *item,
                           const int32_t *rating, int32_t target_chunks,
                           mmsbm_hip_layout **out);
int mmsbm_hip_layout_array(const mmsbm_hip_layout *layout, int which, int32_t *out,
                           int64_t capacity, int64_t *count);
int mmsbm_hip_layout_free(mmsbm_hip_layout *layout);
/* Small problems with uneven degrees (the two-launch iteration): the lists that give every workgroup WHOLE segments
 * -- all pieces of a cut segment -- so that their partial rows are added up in its LDS instead of in a combine
 * launch.  side 0: pair segments, the 64-pair units rebuilt with at most cap_items work items each; side 1: user
 * segments, workgroups of at most cap_items items (a longer segment gets a workgroup of its own).  which: 0 units
 * (items begin, end, splits begin, end), 1 work items in (segment, piece) order (segment, begin, end, workgroup-local
 * partial row or -1), 2 split segments (segment, first local partial row, pieces, 1 = combined in the strided order of
 * seg_combine_kernel), 3 the rebuilt unit list as (rating, q_begin, q_end, 0) (side 0), 4 { most partial rows a
 * workgroup holds, 1 if the lists could be built }.  Pass out == NULL to query count. */
int mmsbm_hip_layout_fused(const mmsbm_hip_layout *layout, int side, int32_t cap_items, int which, int32_t *out,
                           int64_t capacity, int64_t *count);
/* No exception ever crosses this ABI: every entry point runs inside one handler that turns whatever is thrown
 * -- the library's own errors, std::exception, std::bad_alloc, and anything else -- into a status code and a
 * message for mmsbm_hip_last_error().  This entry throws on purpose from inside that handler so the rule can be
 * tested without a GPU: kind 0 nothing (returns MMSBM_OK), 1 std::invalid_argument (MMSBM_E_INVALID), 2
 * std::runtime_error, 3 std::bad_alloc, 4 an int, 5 a class not derived from std::exception (all four:
 * MMSBM_E_INTERNAL). */
int mmsbm_hip_selftest_throw(int kind);

#ifdef __cplusplus
}
#endif
#endif /* MMSBM_HIP_H */
