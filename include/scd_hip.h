/* scd_hip.h - C ABI of libscd_hip.so: the MI355X (gfx950) hot path of Visual-AI/SCD.
 *
 * The reference has no C ABI: its hot path is Python calling torch/sklearn/OR-Tools
 * (SURVEY.md section 8b).  Each entry point below replaces the reference code cited
 * next to it (paths relative to /root/reference); INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative scd_status; scd_last_error()
 *    returns a thread-local message for the last failure;
 *  - device pointers are raw (tensor.data_ptr()); the library never allocates or frees
 *    caller-visible memory: scratch comes from the `ws` argument whose size the matching
 *    *_ws_bytes() query returns (16-byte aligned);
 *  - all device work is enqueued on `stream` (a hipStream_t passed as void*) and is
 *    asynchronous; no call synchronises the device unless stated;
 *  - one handle per device / rank; handles are not shared between host threads.
 */
#ifndef SCD_HIP_H
#define SCD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scd_ctx* scd_handle;
typedef struct scd_encoder scd_encoder;

enum scd_status { SCD_OK = 0, SCD_EINVAL = -1, SCD_EHIP = -2, SCD_ERCCL = -3, SCD_EINFEASIBLE = -4 };
enum scd_dtype { SCD_F32 = 0, SCD_F16 = 1 };
enum scd_sim_mode { SCD_SIM_RAW = 0, SCD_SIM_SOFTMAX = 1 };

int scd_version(void);
const char* scd_last_error(void);
int scd_create(int device, scd_handle* out);
int scd_destroy(scd_handle h);
/* Profiling aid: launches an empty kernel named scd_mark_begin_kernel (end = 0) or scd_mark_end_kernel (end != 0) on `stream`, so that a
 * kernel trace shows where a measured region starts and ends (bench.py brackets its timed steps; tools/trace_window_stats.py). */
int scd_trace_mark(scd_handle h, int end, void* stream);

/* ---- L2 normalisation: F.normalize(feats, dim=-1) main_unsup.py:130; clip_lang_util.py:103-105 ---- */
int scd_l2norm_rows(scd_handle h, const void* x, int dtype, int64_t n, int d, void* out, void* stream);

/* ---- similarity + top-k: main_unsup.py:504-531, main_ptsup.py:526-545 (a5); argmax re-classification
 *      main_unsup.py:601-614, main_ptsup.py:668-676, get_clip_preds_fast main_ptsup.py:78-99 (a6).
 * F  [n,d] fp16 row-major (image features);  Wt [v,d] fp16 row-major = zeroshot_weights.T (name-major).
 * logits = scale * F @ Wt^T; order = (value desc, index asc) on the exact (float64) dot products.
 * idx_out int64 [n,k]; val_out float32 [n,k] (softmax probability when mode == SCD_SIM_SOFTMAX). k <= 8.
 * fallback_rows_out (device int32, may be NULL) counts rows that took the exact full-row path. */
size_t scd_sim_topk_ws_bytes(int64_t n, int d, int64_t v, int k);
int scd_sim_topk(scd_handle h, const void* F, const void* Wt, int64_t n, int d, int64_t v, float scale, int k,
                 int mode, int64_t* idx_out, float* val_out, int32_t* fallback_rows_out, void* ws, size_t ws_bytes,
                 void* stream);
/* The vocabulary is constant over a run (main_unsup.py:389-394 loads it once; :504-531 uses it for every block of rows): its only
 * data-dependent ingredient in the error bound, max_v ||w_v||^2, can be computed ONCE (wmax2_out: 4 bytes of device memory) and handed
 * to every later call instead of being recomputed by each (12 us of a 2.7-ms call at V = 21,000).  The caller vouches that Wt has not
 * changed since; scd_sim_topk computes the norm itself. */
int scd_sim_vocab_norm(scd_handle h, const void* Wt, int64_t v, int d, void* wmax2_out, void* stream);
int scd_sim_topk_prenorm(scd_handle h, const void* F, const void* Wt, int64_t n, int d, int64_t v, float scale, int k,
                         int mode, int64_t* idx_out, float* val_out, int32_t* fallback_rows_out, void* ws, size_t ws_bytes,
                         const void* wmax2, void* stream);
/* argmax re-classification over the K candidate names (main_unsup.py:601-614, main_ptsup.py:668-676, get_clip_preds_fast
 * main_ptsup.py:78-99): idx_out int64 [n], val_out float32 [n] = scd_sim_topk with k = 1 on the raw logits (ws: scd_sim_topk_ws_bytes(n, d, v, 1)). */
int scd_sim_argmax(scd_handle h, const void* F, const void* Wt, int64_t n, int d, int64_t v, float scale, int64_t* idx_out,
                   float* val_out, void* ws, size_t ws_bytes, void* stream);
/* W [r,c] fp16 -> Wt [c,r]  (zeroshot_weights [512,V] -> name-major) */
int scd_transpose_f16(scd_handle h, const void* in, int64_t r, int64_t c, void* out, void* stream);
/* out[i,:] = Wt[idx[i],:]  (the `zeroshot_weights[:, nouns.index(n)]` gather, main_unsup.py:601-602) */
int scd_gather_rows_f16(scd_handle h, const void* Wt, const int64_t* idx, int64_t m, int d, void* out, void* stream);

/* The row selections of main_unsup.py:318-321,561 (`all_feats[~mask_lab]`, `all_feats[mask_lab]`, `name_idx_top5[~mask_lab]`: numpy /
 * torch fancy indexing in the reference) in one launch: for the m rows idx[i] of F (fp16 [n,d]) out16[i] = the row, out32[i] = its
 * float32 image (the K-Means input), nidx_out[i] = name_idx[idx[i]] (int64 rows of k); any of the three outputs may be NULL. */
int scd_select_rows(scd_handle h, const void* F, const int64_t* name_idx, const int64_t* idx, int64_t m, int d, int k, void* out16,
                    float* out32, int64_t* nidx_out, void* stream);

/* out = fp16((a + b) / 2) over n_elems fp16 values (n_elems % 8 == 0): the textual-enhancement feature of BASELINE configs[4],
 * `100 * (f @ W + t @ W) / 2` (commented at main_unsup.py:518,523,604,609) = 100 * mean(f, t) @ W -> scd_sim_topk on the mean. */
int scd_mean2_f16(scd_handle h, const void* a, const void* b, int64_t n_elems, void* out, void* stream);

/* zeroshot_classifier pooling (local_utils/clip_lang_util.py:103-107): emb fp16 [n_names*t_per, d] prompt embeddings ->
 * per name normalise, mean, normalise; written as columns col0.. of out fp16 [d, ld_out] (torch.stack(dim=1) layout). */
int scd_prompt_pool(scd_handle h, const void* emb, int n_names, int t_per, int d, int64_t col0, int64_t ld_out, void* out,
                    void* stream);

/* ---- K-Means: local_utils/sskm_constrained.py, gcd/methods/clustering/faster_mix_k_means_pytorch.py ---- */
/* one-off per data set: centred, power-of-two scaled fp16 copy of X and its row norms (E-step operand).  The copy carries
 * 32 rows of padding behind row n-1 (the streaming E-step reads whole 32-row units); prep must hold
 * scd_kmeans_prep_bytes(n, d) bytes.  X and C must be 16-byte aligned (float4 loads when d % 4 == 0). */
size_t scd_kmeans_prep_bytes(int64_t n, int d);
int scd_kmeans_prepare(scd_handle h, const float* X, int64_t n, int d, void* prep, void* stream);
/* E-step: labels[i] = argmin_k ||x_i - c_k||^2, ties -> lowest k, decided on float64 values
 * (torch.min(dist,1) faster_mix_k_means_pytorch.py:140,192).  refine_rows_out (device int32, may be NULL)
 * receives the number of rows re-evaluated exactly.  D <= 768 and K <= 2048 take the streaming filter (centre prep, one
 * filter launch per 128 centres, refine), larger shapes the tiled one; the result is the same by construction. */
size_t scd_kmeans_estep_ws_bytes(int64_t n, int d, int k);
int scd_kmeans_estep(scd_handle h, const float* X, const void* prep, const float* C, int64_t n, int d, int k,
                     int32_t* labels_out, int32_t* refine_rows_out, void* ws, size_t ws_bytes, void* stream);
/* Hints for the NEXT scd_kmeans_estep on this handle (one-shot: that call consumes them whichever path it takes).  Results are
 * identical with or without them.
 *   SCD_ESTEP_FEW                     few rows are expected inside the filter's error bound (Lloyd iterations after the first two;
 *                                     converged centres): they are re-evaluated in the tail of the filter kernel instead of by a
 *                                     refine launch.  A wrong hint only costs time.
 *   SCD_ESTEP_CENTRES_FROM_FINALIZE   the caller vouches that the centres it will pass are the C_out of the last
 *                                     scd_kmeans_finalize on this handle (same buffer, NOT modified since, same E-step workspace):
 *                                     that call has already written the E-step's centre operands and the prep launch is skipped.
 *                                     Without the flag a matching pointer is not trusted (the address may have been recycled). */
/* Measurement aid (bench.py): while enabled, the streaming filter launches of every scd_kmeans_estep on this handle - stand-alone
 * or inside scd_kmeans_lloyd_step - are bracketed by HIP events on the launch stream (one pair per call; K > 128: the call's
 * kp / 128 launches together).  Each call returns and clears what was collected: the calls' durations in milliseconds, in call
 * order (the first `cap` of them), and their number; it synchronises on the recorded events. */
int scd_kmeans_timing(scd_handle h, int enable, double* samples_ms_out, int cap, int* launches_out);
#define SCD_ESTEP_FEW 1
#define SCD_ESTEP_CENTRES_FROM_FINALIZE 2
int scd_kmeans_estep_hint(scd_handle h, int flags);
/* d2_out[i] = ||x_i - c_{labels[i]}||^2 (float64 sum rounded to float32) */
int scd_kmeans_rowdist(scd_handle h, const float* X, const float* C, const int32_t* labels, int64_t n, int d, int k,
                       float* d2_out, void* stream);
/* pairwise_distance (sskm_constrained.py:189-224): out[n,k] float32 (mode 0: d2, 1: sqrt(d2)); when cost_out != NULL
 * also writes the int32 flow costs round(1000*sqrt(d2)) (:324). */
int scd_kmeans_dist(scd_handle h, const float* X, const float* C, int64_t n, int d, int k, int mode, float* out,
                    int32_t* cost_out, void* stream);
/* M-step partials (sskm_constrained.py:125-128) + inertia of the same labels against C_old (:118-120):
 * sums[k,d] float64, counts[k] int64, inertia[2] float64 = {rows < split, rows >= split}.  Partials are what a
 * multi-GPU caller all-reduces before scd_kmeans_finalize. */
size_t scd_kmeans_mstep_ws_bytes(int64_t n, int d, int k);
int scd_kmeans_mstep(scd_handle h, const float* X, const int32_t* labels, const float* C_old, int64_t n, int d, int k,
                     int64_t split, double* sums, int64_t* counts, double* inertia, void* ws, size_t ws_bytes,
                     void* stream);
/* The same partials from an fp16 copy of X (half the row bytes): valid when every value of X is exactly representable in fp16,
 * as features that left an fp16 encoder are - the float64 sums are then bit-identical.  scd_f16_exact writes the copy
 * (n_elems % 4 == 0) and counts the blocks that saw a value that does not survive the round trip (*inexact_out == 0: exact). */
int scd_f16_exact(scd_handle h, const float* X, int64_t n_elems, void* out16, int32_t* inexact_out, void* stream);
/* The same, also returning max |x| (device float; +inf when a value is infinite).  The incremental M-step's "exact sums" argument
 * (scd_kmeans_lloyd_step_delta) needs rows * max|x| * 2^24 < 2^53 on top of the exact copy: unit-scale features satisfy it by orders
 * of magnitude, fp16 values near 65504 in clusters of 2^13 rows do not - the caller checks n * max|x| < 2^29. */
int scd_f16_exact_max(scd_handle h, const float* X, int64_t n_elems, void* out16, int32_t* inexact_out, float* absmax_out, void* stream);
int scd_kmeans_mstep_f16(scd_handle h, const void* X16, const int32_t* labels, const float* C_old, int64_t n, int d, int k,
                         int64_t split, double* sums, int64_t* counts, double* inertia, void* ws, size_t ws_bytes,
                         void* stream);
/* centres = sums / counts (empty -> NaN, as torch mean of an empty selection); shift_out (device double, may be NULL)
 * = (sum_k ||c_k - c_old_k||_2)^2  (shift_mode 0: sskm_constrained.py:135-136) or sum_k ||c_k - c_old_k||^2 (shift_mode 1:
 * sklearn's center_shift_tot, the `--cluster KM` path of main_unsup.py:362) */
int scd_kmeans_finalize(scd_handle h, const double* sums, const int64_t* counts, int k, int d, const float* C_old,
                        float* C_out, double* shift_out, int shift_mode, const void* prep, void* estep_ws,
                        size_t estep_ws_bytes, int64_t n, void* stream);
/* One whole Lloyd iteration of the unconstrained K-Means (faster_mix_k_means_pytorch.py:187-214) = scd_kmeans_estep on the n_u
 * unlabelled rows (labels_cat[n_cat - n_u ...] written; the first n_cat - n_u entries are the labelled rows' fixed cluster ids) +
 * scd_kmeans_mstep[_f16] over the n_cat rows [labelled ; unlabelled] (X16_cat: their exact fp16 copy, or NULL) + scd_kmeans_finalize
 * with the hand-over of the next E-step's centre operands, behind one call.  stats: device double [4] = {inertia labelled,
 * inertia unlabelled, centre shift, rows the E-step re-evaluated exactly}.  `expect_few`: SCD_ESTEP_* flags for this step's
 * E-step.  C_out must differ from C_in; ws_e / ws_m as for the single calls. */
int scd_kmeans_lloyd_step(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const float* X_cat,
                          const void* X16_cat, int64_t n_cat, int d, int k, int32_t* labels_cat, const float* C_in,
                          float* C_out, double* sums, int64_t* counts, double* stats, int expect_few, void* ws_e,
                          size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream);
/* The same iteration with an INCREMENTAL M-step, for row sets whose exact fp16 copy exists (scd_f16_exact: then the float64
 * cluster sums are exact, hence independent of the order of additions): sums / counts of the previous iteration are updated with
 * the rows whose label changed (labels_prev: the labels sums / counts belong to, updated in place) and the inertia is evaluated
 * from the sums, the centres and the rows' sum of squares (scd_kmeans_sumsq, double-double) - bit-identical centres, labels and
 * float32 inertia at a cost proportional to the changes.  flags: SCD_ESTEP_* | SCD_LLOYD_FULL (a fresh M-step over the rows, as
 * scd_kmeans_lloyd_step; the first iterations of a restart, or whenever many labels move).  sums_lab / counts_lab: sums / counts of
 * the labelled rows alone (NULL when there are none).  stats: device double [5] = {inertia labelled, inertia unlabelled, centre
 * shift, rows re-evaluated exactly, rows whose label changed}.  Reference: faster_mix_k_means_pytorch.py:187-214. */
#define SCD_LLOYD_FULL 8
/* (scd_kmeans_sumsq reads the rows as one flat array with 16-byte loads: the base pointer must be 16-byte aligned - an unaligned one is
 * refused with SCD_EINVAL; n, d and split are free.) */
int scd_kmeans_sumsq(scd_handle h, const void* X16, const float* X, int64_t n, int d, int64_t split, double* out4, void* stream);
int scd_kmeans_lloyd_step_delta(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat,
                                int64_t n_cat, int d, int k, int32_t* labels_cat, int32_t* labels_prev, const float* C_in,
                                float* C_out, double* sums, int64_t* counts, const double* sums_lab, const int64_t* counts_lab,
                                const double* sumsq4, double* stats, int flags, void* ws_e, size_t ws_e_bytes, void* ws_m,
                                size_t ws_m_bytes, void* stream);
/* One restart's whole Lloyd loop behind one call (faster_mix_k_means_pytorch.py:187-214: up to max_iter iterations of
 * scd_kmeans_lloyd_step_delta from C_start, stop after the iteration whose centre shift is below tol, keep the labels / centres of
 * the iteration with the least float32 inertia).  The host stays one iteration behind the device (iteration i + 1 is enqueued
 * before iteration i's statistics are looked at; a speculative iteration behind a converged one is dropped); the statistics reach
 * the host through pinned memory written by the iteration's last kernel, so nothing but the iterations' kernels enters the stream.
 * Caller-owned device rings: lab_ring int32 [3][n_cat] (iteration i's labels in slot i % 3; labels_lab = the n_cat - n_u labelled
 * rows' labels, NULL when there are none), C_ring float [3][k,d], stats_ring double [2][5].  Outputs: best_labels int32 [n_cat] and
 * best_C [k,d] on the device (valid in stream order), result_host (HOST double [4]) = {least inertia, iterations done, iterations
 * with the incremental M-step, iterations launched}.  Returns when the last iteration's statistics have arrived.
 * An iteration whose M-step leaves a cluster without rows ends the restart the way the reference's loop does (:140-160, :192-214: that
 * centre is NaN, `torch.min` then yields NaN at the first NaN column for every row, so no later iteration is ever the best or ends the
 * loop): the best of the iterations so far is kept - its centres carry NaN rows if it is that iteration - and "iterations done" is
 * max_iter.  The one configuration in which the reference's loop can recover (exactly one cluster without labelled rows) is the
 * caller's to route elsewhere (scd_amd/kmeans.py does). */
int scd_kmeans_lloyd_run(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat, int64_t n_cat, int d,
                         int k, const int32_t* labels_lab, int32_t* lab_ring, int32_t* labels_prev, const float* C_start,
                         float* C_ring, double* sums, int64_t* counts, const double* sums_lab, const int64_t* counts_lab,
                         const double* sumsq4, double* stats_ring, int max_iter, double tol, int32_t* best_labels, float* best_C,
                         double* result_host, void* ws_e, size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream);
/* The same loop over a ROW SHARD, one process per GPU (the reference is single-process): X_u / X16_cat / labels are this
 * rank's rows, sums_lab / counts_lab / sumsq4 the GLOBAL ones (reduced by the caller once per fit).  Every iteration packs the rank's
 * [k*d sums | k counts as float64] into xbuf (device, k*d + 2k doubles) and calls exchange(exchange_ctx, xbuf, k*d + k, stream), which
 * must leave the element-wise sum over all ranks in xbuf in stream order (an all-reduce; scd_allreduce_centroids has this signature
 * with ctx = the handle, a torch.distributed caller passes a callback around dist.all_reduce) and return 0.  Centres, centre shift
 * and inertia are then formed from the global sums, so every rank takes the same stop / keep decisions and issues the same number of
 * exchanges; with exact sums (scd_f16_exact_max: GLOBAL rows x max|x| < 2^29) the result is bit-identical to the single-process
 * loop whatever the sharding.  The fresh-or-incremental M-step choice and the E-step hints are per rank.  Needs n_u > 0 on every
 * rank.  A non-zero return of the callback ends the loop with SCD_ERCCL. */
typedef int (*scd_exchange_fn)(void* ctx, double* buf, int64_t n_doubles, void* stream);
int scd_kmeans_lloyd_run_sharded(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat, int64_t n_cat,
                                 int d, int k, const int32_t* labels_lab, int32_t* lab_ring, int32_t* labels_prev, const float* C_start,
                                 float* C_ring, double* sums, int64_t* counts, const double* sums_lab, const int64_t* counts_lab,
                                 const double* sumsq4, double* stats_ring, int max_iter, double tol, int32_t* best_labels, float* best_C,
                                 double* result_host, void* ws_e, size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream,
                                 double* xbuf, scd_exchange_fn exchange, void* exchange_ctx);
/* ALL restarts of a fit in lock-step (faster_mix_k_means_pytorch.py:244-275: the n_init restarts are independent once seeded): iteration
 * i of every restart still running is enqueued, then iteration i - 1 of each is settled; a restart that has converged drops out.  Every
 * restart executes exactly the launches of scd_kmeans_lloyd_run with the same arguments, so its outputs are the same bits.  restarts[j]
 * = restart j's own handle (scratch, centre hand-over and statistics ring are per handle: R distinct handles of the current device) and
 * buffers - the per-restart arguments of scd_kmeans_lloyd_run, with E-step / M-step workspaces of its own; the rest is shared.
 * xbuf / exchange NULL: one process.  Otherwise a row shard as in scd_kmeans_lloyd_run_sharded, with ONE exchange per iteration for all
 * restarts: xbuf holds R * (k*d + 2k) doubles, the call is exchange(ctx, xbuf, (restarts still running) * (k*d + k), stream) with the
 * running restarts' [k*d sums | k counts] packed densely in restart order (the same set on every rank: the stop decisions come from
 * exchanged statistics) - R times fewer collectives per fit than one run per restart.
 * n_streams > 1: the restarts' launches are spread over that many library-owned streams, ordered behind what `stream` already holds;
 * `stream` continues behind all of them (under a group the exchange itself stays on `stream`). */
typedef struct scd_lloyd_restart {
    scd_handle h;
    int32_t* lab_ring;        /* [3][n_cat] */
    int32_t* labels_prev;     /* [n_cat] */
    const float* C_start;     /* [k][d] the seeding */
    float* C_ring;            /* [3][k][d] */
    double* sums;             /* [k][d] */
    int64_t* counts;          /* [k] */
    double* stats_ring;       /* [2][5] */
    int32_t* best_labels;     /* out [n_cat] */
    float* best_C;            /* out [k][d] */
    double* result_host;      /* out, host [4]: float32 inertia, iterations done, incremental iterations, iterations launched */
    void* ws_e;               /* scd_kmeans_estep_ws_bytes */
    void* ws_m;               /* scd_kmeans_mstep_ws_bytes */
} scd_lloyd_restart;
int scd_kmeans_lloyd_run_multi(const scd_lloyd_restart* restarts, int R, const float* X_u, const void* prep_u, int64_t n_u,
                               const void* X16_cat, int64_t n_cat, int d, int k, const int32_t* labels_lab, const double* sums_lab,
                               const int64_t* counts_lab, const double* sumsq4, int max_iter, double tol, size_t ws_e_bytes,
                               size_t ws_m_bytes, void* stream, double* xbuf, scd_exchange_fn exchange, void* exchange_ctx, int n_streams);

/* The lock-step k-means++ rounds of scd_kpp_seed_lockstep over a ROW SHARD (one process per GPU; sskm_constrained.py:28-44 is
 * single-process): per round three exchanges - shard sums, shard probability masses, candidate rows - each an all-gather of a few bytes
 * per restart through `gather(gather_ctx, send, recv, bytes_per_rank, stream)`: rank w's `bytes_per_rank` bytes at `send` must arrive at
 * recv + w * bytes_per_rank on every rank, in stream order; send / recv lie inside xbuf (device, scd_kpp_seed_sharded_xbuf_bytes).  The
 * new centres are the rows of their first owner in rank order (float32 values of the global X), written to C_buf[r][m0 + t]; picks_out
 * int64 [T][R] = 0, or -1 where no shard reported a hit (the reference indexes an empty nonzero() there).  d2 float [R][ld]: this
 * shard's closest squared distances, updated in place; r_dev float [T][R]: the restarts' uniforms; ws: scd_kpp_seed_sharded_ws_bytes.
 * X16 (may be NULL): this shard's exact fp16 copy - the update then goes through the MFMA filter as in scd_kpp_seed_lockstep.
 * Every rank needs at least one row (n > 0) and must make the call (three gathers per round, T rounds).  A non-zero return of the
 * callback ends the loop with SCD_ERCCL. */
typedef int (*scd_gather_fn)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
size_t scd_kpp_seed_sharded_ws_bytes(int64_t n, int d, int R);
size_t scd_kpp_seed_sharded_xbuf_bytes(int d, int R, int world);
int scd_kpp_seed_lockstep_sharded(scd_handle h, const float* X, const void* X16, int64_t n, int d, int R, float* d2, int64_t ld,
                                  const float* r_dev, int T, float* C_buf, int k, int m0, int64_t* picks_out, void* ws, size_t ws_bytes,
                                  void* stream, void* xbuf, size_t xbuf_bytes, scd_gather_fn gather, void* gather_ctx, int rank, int world);
/* prep / estep_ws (both may be NULL): the data set's scd_kmeans_prepare buffer and the workspace the NEXT scd_kmeans_estep
 * of these centres will be given (n = its row count).  The blocks that produce the centres then also write their E-step
 * operands into it, and that E-step - same handle, same C_out pointer, same workspace, C_out unmodified in between - skips
 * its centre-prep launch.  Any other scd_kmeans_finalize / scd_kmeans_estep call on the handle drops the hand-over. */
/* sklearn.cluster.KMeans pieces (main_unsup.py:362, main_ptsup.py:381 `KMeans(n_clusters, random_state=0).fit(u_feats)`):
 * out (device int64) = number of rows whose label changed (strict convergence, `np.array_equal(labels, labels_old)`); */
int scd_labels_changed(scd_handle h, const int32_t* a, const int32_t* b, int64_t n, int64_t* out, void* stream);
/* the greedy k-means++ candidate draw: idx_out[l] = searchsorted(cumsum_f64(d2), u[l] * float32(sum d2)) clipped to n-1 for
 * l < n_draws (u: device doubles, the host RandomState's uniforms); pot_out (device double, may be NULL) = sum d2.
 * ws: scd_kpp_draw_ws_bytes(n). */
int scd_kpp_searchsorted(scd_handle h, const float* d2, int64_t n, const double* u, int n_draws, int64_t* idx_out,
                         double* pot_out, void* ws, size_t ws_bytes, void* stream);
/* The same greedy seeding for the R = n_init starts of a fit in lock-step, the rounds in C (the starts share X; each owns a fixed
 * slice of the host RandomState's stream: its first centre, then L = 2 + int(ln k) uniforms per added centre - `_k_init` of the
 * scikit-learn the reference vendors, local_utils/k_means_constrained/sklearn_import/cluster/k_means_.py:33-132, = `_kmeans_plusplus`
 * of the pinned 1.0.2).  first: device int64 [R] rows of the first centres; u: device double [k-1][R][L]; C_buf float [R][k][d] and
 * picks_out int64 [k][R] receive the centres and their row indices.  X16: the exact fp16 copy of X (scd_f16_exact) or NULL: with it
 * the R * L candidates of a round are measured through the MFMA lower-bound filter of scd_kpp_seed_lockstep, without it densely -
 * the same potentials (float64 sums of float32(float64-exact) distances), the same picks. */
size_t scd_kpp_greedy_ws_bytes(int64_t n, int d, int R, int L);
int scd_kpp_greedy_lockstep(scd_handle h, const float* X, const void* X16, int64_t n, int d, int R, int L, int k, const int64_t* first,
                            const double* u, float* C_buf, int64_t* picks_out, void* ws, size_t ws_bytes, void* stream);
/* sklearn's `_kmeans_single_lloyd` behind one call, for rows with an exact fp16 copy (the Lloyd loop of `KMeans.fit`, main_unsup.py:362):
 * iterations of scd_kmeans_lloyd_step_delta with sklearn's centre shift (sum_k ||dc_k||^2) from C_start until no label changes
 * (never at the first iteration) or the shift is <= tol or max_iter is reached, then the E-step of the final centres.  The host stays
 * one iteration behind the device as in scd_kmeans_lloyd_run (same rings: lab_ring int32 [3][n], C_ring float [3][k,d], stats_ring
 * double [2][5]; labels_prev int32 [n]).  final_labels int32 [n], final_C [k,d] (device, valid in stream order).  result_host (HOST
 * double [4]) = {status, n_iter, iterations with the incremental M-step, iterations launched}; status 1: an iteration left a cluster
 * empty - sklearn re-seeds it with a far point (`_relocate_empty_clusters_dense`), the caller runs that start itself. */
int scd_kmeans_lloyd_run_sk(scd_handle h, const float* X, const void* prep, int64_t n, const void* X16, int d, int k, int32_t* lab_ring,
                            int32_t* labels_prev, const float* C_start, float* C_ring, double* sums, int64_t* counts,
                            const double* sumsq4, double* stats_ring, int max_iter, double tol, int32_t* final_labels, float* final_C,
                            double* result_host, void* ws_e, size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream);
/* incremental k-means++ (kpp, sskm_constrained.py:28-44): d2 = min(d2, ||x - c_new||^2) */
int scd_kmeans_min_update(scd_handle h, const float* X, const float* c_new, int64_t n, int d, float* d2_inout,
                          void* stream);
/* one draw: prob = d2/float32(total); first i with float32(prefix + cumsum_f64(prob))[i] >= r; idx_out device int64
 * (-1: none on this shard).  total / prefix (device doubles, may be NULL = local sum / 0) make the draw shard-aware for
 * multi-GPU k-means++; probsum_out (device double, may be NULL) receives this shard's sum of prob. */
size_t scd_kpp_draw_ws_bytes(int64_t n);
int scd_kpp_draw(scd_handle h, const float* d2, int64_t n, float r, const double* total, const double* prefix,
                 int64_t* idx_out, double* probsum_out, void* ws, size_t ws_bytes, void* stream);
/* The same for R restarts in lock-step (the n_init restarts of one fit share X and a fixed random stream): d2 [R][ld] (ld >= n),
 * c_new [R][d], r_dev device float [R], total / prefix / idx_out / probsum_out device arrays of R (or NULL as above).  X is read
 * once per round instead of R times; ws: R * scd_kpp_draw_ws_bytes(n).  scd_sum_f32_multi: out[r] = float64 sum of row r. */
int scd_kmeans_min_update_multi(scd_handle h, const float* X, const float* c_new, int64_t n, int d, int R, float* d2_inout,
                                int64_t ld, void* stream);
int scd_kpp_draw_multi(scd_handle h, const float* d2, int64_t n, int64_t ld, int R, const float* r_dev, const double* total,
                       const double* prefix, int64_t* idx_out, double* probsum_out, void* ws, size_t ws_bytes, void* stream);
int scd_sum_f32_multi(scd_handle h, const float* x, int64_t n, int64_t ld, int R, double* out, void* stream);
/* The rounds of the lock-step seeding behind one call (kpp of sskm_constrained.py:28-44 for R restarts at once): for t < T: draw
 * one row per restart from d2 with the uniforms r_dev[t][R] (scd_kpp_draw_multi), store it as centre m0 + t of C_buf [R][k][d],
 * d2[r] = min(d2[r], ||x - that row||^2).  On entry d2 [R][ld] holds the distances to the m0 centres chosen so far.  picks_out
 * int64 [T][R] on the device (-1: no row drawn; the reference raises there).  X16: the exact fp16 copy of X (scd_f16_exact) or
 * NULL: with it the distance update reads 2 bytes per value through a filter (16x16x32 MFMA lower bounds rule out the rows a new
 * centre cannot improve, the rest get the float64 value) - the same float32 results as scd_kmeans_min_update_multi. */
size_t scd_kpp_seed_ws_bytes(int64_t n, int d, int R);
int scd_kpp_seed_lockstep(scd_handle h, const float* X, const void* X16, int64_t n, int d, int R, float* d2, int64_t ld,
                          const float* r_dev, int T, float* C_buf, int k, int m0, int64_t* picks_out, void* ws, size_t ws_bytes,
                          void* stream);
/* One round's distance update of the lock-step seeding through that filter, as a call of its own: d2[r] = min(d2[r], ||x - c_new[r]||^2),
 * r < R <= 16, c_new [R][d] float32 (rows of the global X), X16 = the exact fp16 copy of this rank's rows.  Same float32 results as
 * scd_kmeans_min_update_multi.  For callers whose rounds cannot run behind scd_kpp_seed_lockstep: under a process group three
 * all-gathers sit between a round's draw and its update.  ws keeps the rows' norm table between calls:
 * first_call != 0 (re)builds it - pass it on the first update of a seeding. */
size_t scd_kpp_update_ws_bytes(int64_t n, int d);
int scd_kpp_update_filter(scd_handle h, const void* X16, int64_t n, int d, int R, const float* c_new, float* d2, int64_t ld,
                          int first_call, void* ws, size_t ws_bytes, void* stream);
/* deterministic float64 sum of a float32 vector (inertia, d2.sum()) */
int scd_sum_f32(scd_handle h, const float* x, int64_t n, double* out, void* stream);

/* ---- vote histogram: Counter(...).most_common(m) per cluster, main_unsup.py:573-586, main_ptsup.py:636-648.
 * name_idx int64 [n, ld] (first top_k columns used); preds int64 [n]; clusters int64 [n_clusters] (ids to vote for);
 * known int64 [n_known] vocabulary ids to drop (may be NULL).  Output per cluster c: keys_out[c, 0..m) and
 * counts_out[c, 0..m) in most_common order (count desc, first-seen asc), padded with -1 / 0. */
size_t scd_vote_hist_ws_bytes(int64_t n, int top_k);
int scd_vote_hist(scd_handle h, const int64_t* name_idx, int64_t n, int ld, int top_k, const int64_t* preds,
                  const int64_t* clusters, int n_clusters, const int64_t* known, int n_known, int m,
                  int64_t* keys_out, int32_t* counts_out, void* ws, size_t ws_bytes, void* stream);
/* The histogram over row SHARDS (one process per GPU).  scd_vote_table: this rank's dense tables counts int32 [C, V]
 * (occurrences of a name among the first top_k columns of the rows predicted as cluster c) and first int64 [C, V] (smallest
 * (row_offset + row) * top_k + column at which it occurs; 0x7f7f7f7f7f7f7f7f = never) - slot_of int32 [n_slots] maps a prediction to its
 * table row (-1: not voted on).  The caller all-reduces counts with SUM and first with MIN over the ranks;
 * scd_vote_table_topm then reads most_common(m) of every cluster off the reduced tables, (count desc, first asc) - the order of
 * Counter.most_common on the concatenated rows (main_unsup.py:573-586).  It zeroes the counts it takes. */
int scd_vote_table(scd_handle h, const int64_t* name_idx, int64_t n, int ld, int top_k, const int64_t* preds, const int32_t* slot_of,
                   int n_slots, int64_t row_offset, int64_t v, int n_clusters, int32_t* counts, int64_t* first, void* stream);
int scd_vote_table_topm(scd_handle h, int32_t* counts, const int64_t* first, int n_clusters, int64_t v, int m, int64_t* keys_out,
                        int32_t* counts_out, void* stream);

/* ---- host solvers (CPU, synchronous) ---- */
/* linear_assignment (gcd/project_utils/cluster_utils.py:234-493), same tie-breaking; pairs_out [min(n,m),2] sorted */
int scd_munkres(const int64_t* cost, int n, int m, int64_t* pairs_out, int* n_pairs_out);
/* assign_name's solve (local_utils/clip_lang_util.py:167-178): linear_assignment(w.max() - w) for the d x d vote matrix w
 * given by its nnz non-zero entries (rows/cols int32, vals int64; duplicates add up as `w[i, col] += v` does) - the same
 * decisions as scd_munkres on the dense matrix, in O(d) per covered row instead of O(d^2): d reaches 10,000-20,000 at
 * K = 1000 clusters (BASELINE configs[3]).  pairs_out [d,2] sorted by row. */
int scd_munkres_sparse(int d, int64_t nnz, const int32_t* rows, const int32_t* cols, const int64_t* vals,
                       int64_t* pairs_out, int* n_pairs_out);
/* solve_min_cost_flow_graph (sskm_constrained.py:331-356) on the transportation form: cost int32 [n,k] */
int scd_transport_solve(const int32_t* cost, int64_t n, int k, int size_min, int size_max, int32_t* labels_out,
                        int64_t* total_cost_out);
/* `batch` problems of one shape - the E-steps of a fit's restarts (sskm_constrained.py:165-176, independent once seeded) - on up to
 * `threads` host threads: cost int32 [batch,n,k], labels_out int32 [batch,n], totals_out int64 [batch] (or NULL).  Every problem's
 * result is scd_transport_solve's for it, whatever the thread count; the status is that of the first problem that failed. */
int scd_transport_solve_batch(const int32_t* cost, int64_t n, int k, int batch, int size_min, int size_max, int32_t* labels_out,
                              int64_t* totals_out, int threads);

/* ---- multi-GPU exchanges over RCCL (one process per GPU, one communicator per handle).  The reference is a
 * single process; these are the two collectives the sharded hot path needs.  librccl is dlopen'ed by the first call.
 *   rank 0: scd_comm_unique_id(id) -> ship the scd_comm_unique_id_bytes() bytes to every rank (file, socket, MPI ...)
 *   all   : scd_comm_init(h, rank, world, id) ... scd_comm_destroy(h)
 * scd_allreduce_centroids: in-place float64 sum of the packed M-step partials [sums k*d | counts k | inertia 2] (what
 *   scd_kmeans_mstep produces, counts converted to float64) - ONE collective per Lloyd iteration;
 * scd_allgather_text: w_full [world * shard_elems] fp16 <- every rank's w_shard (name-major classifier rows, equal shards). */
size_t scd_comm_unique_id_bytes(void);
int scd_comm_unique_id(void* id_out);
int scd_comm_init(scd_handle h, int rank, int world, const void* unique_id);
int scd_comm_destroy(scd_handle h);
int scd_allreduce_centroids(scd_handle h, double* packed, int64_t count, void* stream);
int scd_allgather_text(scd_handle h, const void* w_shard, int64_t shard_elems, void* w_full, void* stream);

/* ---- encoders: CLIP ViT-B/16 visual / text (third-party `clip`, call sites main_unsup.py:127,
 *      clip_lang_util.py:101-102) and DINO ViT-B/16 (gcd/models/vision_transformer.py:135-219) ---- */
typedef struct scd_encoder_desc {
    int kind;          /* 0 CLIP visual, 1 CLIP text, 2 DINO/GCD ViT */
    int width, layers, heads, mlp_dim;
    int tokens;        /* 197 / 77 */
    int patch, image;  /* 16, 224 (visual) */
    int vocab;         /* 49408 (text) */
    int out_dim;       /* 512 (CLIP) or 0 = no projection (DINO) */
    int act;           /* 0 QuickGELU, 1 erf GELU */
    float ln_eps;
} scd_encoder_desc;
/* weights: array of device pointers in the order documented in DESIGN.md (fp16 matrices, fp32 vectors). */
int scd_encoder_create(scd_handle h, const scd_encoder_desc* desc, const void* const* weights, int n_weights,
                       scd_encoder** out);
int scd_encoder_destroy(scd_encoder* e);
size_t scd_encoder_ws_bytes(const scd_encoder* e, int batch);
/* Measurement aid: while enabled, every fc1 GEMM launch of the encoder (its dominant kernel) is bracketed by HIP events on
 * the launch stream.  Each call returns and clears what was collected so far: total milliseconds, launches, algorithmic
 * FLOPs (2*M*N*K per launch); it synchronises on the recorded events. */
int scd_encoder_timing(scd_encoder* e, int enable, double* ms_out, int* launches_out, double* flop_out);
/* pixels [B,3,H,W] (dtype f32/f16) -> out fp16 [B,out] (L2-normalised when normalize != 0) */
int scd_vit_encode_image(scd_handle h, const scd_encoder* e, const void* pixels, int dtype, int batch, void* out,
                         int normalize, void* ws, size_t ws_bytes, void* stream);
/* tokens int32 [B,77] -> out fp16 [B,out] */
int scd_clip_encode_text(scd_handle h, const scd_encoder* e, const int32_t* tokens, int batch, void* out,
                         int normalize, void* ws, size_t ws_bytes, void* stream);
/* The same with only the first ctx_len positions of every prompt computed (tokens stays [B,77]).  Precondition: ctx_len >
 * the EOT position (argmax over the 77 ids) of every row - then the outputs are bit-identical to scd_clip_encode_text (the tower is
 * causal and only the EOT position is read, reference clip model.py encode_text), at ctx_len/77 of the work. */
int scd_clip_encode_text_len(scd_handle h, const scd_encoder* e, const int32_t* tokens, int batch, int ctx_len, void* out,
                             int normalize, void* ws, size_t ws_bytes, void* stream);
/* building block exposed for tests: C[m,n] = A[m,k] @ W[n,k]^T (+bias)(act)(+residual), fp16 in/out, fp32 accumulate */
int scd_gemm_f16(scd_handle h, const void* A, const void* W, const float* bias, const void* residual, void* C,
                 int64_t m, int n, int k, int act, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCD_HIP_H */
