/* plaidhip.h -- C ABI of the MI355X-native gene-set scoring hot path.
 *
 * Drop-in boundary for the R package bigomics/plaid (reference @ 2025-06-14).  The
 * reference has NO native interface (NAMESPACE:1-16 has no useDynLib, there is no src/);
 * the seam is therefore inside the R functions named below, whose BODIES are replaced by
 * `.Call()` into this library while their R signatures stay (see INTEGRATION.md and
 * r-pkg/).  Every entry point cites the reference code it replaces.
 *
 * Conventions
 *   - plain C, no R / torch / HIP types in any signature; `void*` device pointers.
 *   - matrices use R layout: column-major `double`; sparse = dgCMatrix slots
 *     (`p` int32[ncol+1], `i` int32[nnz] 0-based sorted, `x` double[nnz]).
 *   - every function returns a status code (0 = ok); the text of the last error of the
 *     calling thread is available from plaidhip_last_error_string().  Nothing throws.
 *   - element offsets are 64-bit: m*n may exceed 2^31-1 (the reason R/plaid.R:103-104
 *     chunks).
 *   - `host` entry points take caller-owned host buffers, stage them through HBM and
 *     synchronise before returning (R's .Call contract).  `dev` entry points take
 *     device pointers, enqueue on the context's stream and do NOT synchronise.
 */
#ifndef PLAIDHIP_H
#define PLAIDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLAIDHIP_VERSION 200 /* 0.2.0 */

enum plaidhip_status {
  PLAIDHIP_OK = 0,
  PLAIDHIP_EINVAL = 1,       /* bad dimensions / arguments (R: stop())            */
  PLAIDHIP_ENOMEM = 2,       /* device or host allocation failed                  */
  PLAIDHIP_EHIP = 3,         /* a HIP runtime call or kernel failed               */
  PLAIDHIP_EUNSUPPORTED = 4, /* shape outside what the kernels cover              */
  PLAIDHIP_ENODEVICE = 5     /* no gfx950 device visible                          */
};

enum plaidhip_stat { PLAIDHIP_STAT_MEAN = 0, PLAIDHIP_STAT_SUM = 1 }; /* R/plaid.R:60 `stats` */
enum plaidhip_ties {                                                   /* R/plaid.R:593 `ties.method` */
  PLAIDHIP_TIES_AVERAGE = 0,
  PLAIDHIP_TIES_MIN = 1,
  PLAIDHIP_TIES_MAX = 2,
  /* passed through like the reference does (R/plaid.R:614-617 -> matrixStats::colRanks; :639-642 -> base::rank): ties in
   * order of their position / reverse position / without gaps ("dense": dense columns only, as in matrixStats).  Composed
   * from two or three passes of the min-rank kernels: exact, off the hot path, no fused power / column maximum, and the
   * CSC form reads Xp[n] back (not stream-ordered).                                                                     */
  PLAIDHIP_TIES_FIRST = 3,
  PLAIDHIP_TIES_LAST = 4,
  PLAIDHIP_TIES_DENSE = 5,
  PLAIDHIP_TIES_RANDOM = 6 /* legal in R, REFUSED here (PLAIDHIP_EUNSUPPORTED): not a function of the input */
};
enum plaidhip_ignore_zero { /* R/plaid.R:554 `ignore.zero`: NULL / FALSE / TRUE */
  PLAIDHIP_IGNORE_ZERO_AUTO = -1,
  PLAIDHIP_IGNORE_ZERO_FALSE = 0,
  PLAIDHIP_IGNORE_ZERO_TRUE = 1
};

/* `flags` arguments are device arrays of 4 uint32 words set to 0/1 by the SpMM epilogue /
 * plaidhip_dev_minflags (the caller zeroes them first): [0] a value < 0 was seen, [1] an exact
 * zero, [2] a NaN, [3] reserved (never written by the product library).  0/1 words so that a sample-sharded host can all-reduce(MAX)
 * them in place.  min(x, na.rm=TRUE) == 0  <=>  flags[1] && !flags[0]   (R/plaid.R:556-557).
 * The PLAIDHIP_FLAG_* bits are the in-kernel encoding (bit b <-> word b).                    */
#define PLAIDHIP_FLAG_HAS_NEG 1u
#define PLAIDHIP_FLAG_HAS_ZERO 2u
#define PLAIDHIP_FLAG_HAS_NAN 4u

typedef struct plaidhip_ctx plaidhip_ctx;         /* device, stream, workspace           */
typedef struct plaidhip_geneset plaidhip_geneset; /* device-resident prepared membership */

/* ---- lifecycle --------------------------------------------------------------------- */
int plaidhip_version(void);
const char* plaidhip_last_error_string(void);
int plaidhip_device_count(int* count);
/* `stream`: an existing hipStream_t to enqueue on (e.g. the host framework's current
 * stream) or NULL to create a private one. */
int plaidhip_init(int device, void* stream, plaidhip_ctx** out);
int plaidhip_finalize(plaidhip_ctx* ctx);
int plaidhip_synchronize(plaidhip_ctx* ctx);
/* Precision of the dense crossprod.  PLAIDHIP_PRECISION_F64 (default): fp64 storage and accumulation,
 * scores agree with the reference to ~1e-15.  PLAIDHIP_PRECISION_MIXED (opt-in): the sample columns are
 * staged as fp32 (2^-24 relative rounding of the inputs, ~6e-8 on the scores, inside the 1e-5 bar),
 * sums stay fp64; applies to dense X with 8,192 < genes <= 20,448, everything else keeps fp64.        */
enum { PLAIDHIP_PRECISION_F64 = 0, PLAIDHIP_PRECISION_MIXED = 1 };
int plaidhip_set_precision(plaidhip_ctx* ctx, int mode);
/* Enqueue on `stream` (a hipStream_t) from now on; NULL means the device's null stream, e.g. the
 * default stream of a host framework (plaidhip_init treats NULL as "create a private stream", so a
 * caller that wants the null stream says so here).  A private stream created by init is destroyed. */
int plaidhip_set_stream(plaidhip_ctx* ctx, void* stream);
/* Kernel-selection knobs of one context (tests and tools use them to pin a path; the defaults choose
 * by shape).  Unknown option or value: PLAIDHIP_EINVAL.                                              */
enum plaidhip_option {
  PLAIDHIP_OPT_SPMM_DENSE_KERNEL = 1,  /* 0 auto (default) | 1 one-column kernel | 2 pair kernel wherever it applies |
                                          3 dense 0/1 G x bf16x3 split of X on MFMA (BASELINE config 4's "GEMM" form:
                                          ~145x the flops of the SpMM, ~1e-7 relative; measured beside it, never default) */
  PLAIDHIP_OPT_SPMM_SPARSE_KERNEL = 2, /* 0 auto: by nnz(X) (default) | 1 scatter | 2 gather                        */
  PLAIDHIP_OPT_NT_STORE = 3,           /* -1 auto (default) | 0 plain stores of S | 1 streaming stores              */
  PLAIDHIP_OPT_RANKS_F32 = 4,          /* staging of RANK inputs in the crossprod, all three exact and bit-identical:
                                          2 (default) u16 (2 * rank), four samples per LDS entry, integer sums |
                                          1 fp32 staging | 0 the fp64 kernels                                        */
  PLAIDHIP_OPT_RANK_KERNEL = 5,        /* 0 auto (default) | 1 sorting network | 2 bucket ranker | 3 bucket ranker with
                                          512 threads x 40 keys for columns beyond 12,288 keys (default: 1,024 x 20)  */
  PLAIDHIP_OPT_SCATTER_FIXED = 6,      /* sparse-X scatter kernel, inputs declared bounded (rank weights): 1 (default) u64
                                          fixed-point accumulators: exact integer sums, bit-reproducible | 0 fp64 atomics  */
  PLAIDHIP_OPT_SCATTER_ORDER = 7,      /* sparse-X scatter kernel: 1 (default) all workgroups on one chunk of sets at a
                                          time (chunk, column order) | 0 column after column                              */
  PLAIDHIP_OPT_FUSED_MEDIANS = 8       /* plaidhip_dev_spmm_csc_fused_f64 and the host pipelines on a dgCMatrix: medians
                                          selected inside the crossprod launch 0 (default) from 1e9 scores on | 1 whenever
                                          the shapes allow | 2 never                                                      */
};
int plaidhip_set_option(plaidhip_ctx* ctx, int option, int value);
/* Size limits of the kernels a host has to route by (so that no binding repeats them as literals).  Unknown `which`:
 * PLAIDHIP_EINVAL.                                                                                                   */
enum plaidhip_limit_id {
  PLAIDHIP_LIMIT_SPARSE_RANK_COLUMN = 1, /* most stored values of a column plaidhip_dev_colranks_csc_dense_nz_f64 takes     */
  PLAIDHIP_LIMIT_LDS_GENES = 2           /* most genes the one-slice LDS-resident crossprod kernels take (u16 rank staging) */
};
int plaidhip_limit(int which, int64_t* value);
/* device memory helpers for hosts without a tensor library (R) */
int plaidhip_malloc(plaidhip_ctx* ctx, size_t bytes, void** dptr);
int plaidhip_free(plaidhip_ctx* ctx, void* dptr);
int plaidhip_memcpy_h2d(plaidhip_ctx* ctx, void* dst, const void* src, size_t bytes);
int plaidhip_memcpy_d2h(plaidhip_ctx* ctx, void* dst, const void* src, size_t bytes);

/* ---- gene-set membership G (replaces R/plaid.R:72-77 on `gmt2mat()` output,
 *      R/gmt-utils.R:19-66).  The caller passes the CSC pattern of the ALIGNED,
 *      binarised membership: column j lists the rows OF X (0-based, < g) that belong to
 *      set j, i.e. `matG[gg,] != 0` re-indexed into X's row space (R/plaid.R:65-73) --
 *      X itself is never row-gathered.  Explicit zeros must already be dropped.
 *      Set sizes (colSums(G), R/plaid.R:75) are the column lengths.
 *      A prepared gene-set collection is reused for every sample, chunk and call, but it owns
 *      per-launch device scratch: use it from ONE stream at a time (one per context is free). */
int plaidhip_geneset_create(plaidhip_ctx* ctx, int32_t g, int32_t m, const int32_t* Gp,
                            const int32_t* Gi, plaidhip_geneset** out);
int plaidhip_geneset_destroy(plaidhip_geneset* gs);
/* info[0]=g info[1]=m info[2]=z (nnz) info[3]=padded index slots info[4]=tiles
 * info[5]=1 if the LDS-resident column kernel is usable for f64 at this g             */
int plaidhip_geneset_info(const plaidhip_geneset* gs, int64_t info[8]);

/* ---- device-level hot path (pointers are device pointers) --------------------------- */

/* S = alpha * (G^T X) (.) w + beta * (k (.) w):  the crossprod of R/plaid.R:80,107 with
 * the column scaling of R/plaid.R:74-77 folded into the epilogue.  w_j = 1/(1e-8 + k_j)
 * for STAT_MEAN, 1 for STAT_SUM; k_j = size of set j.  alpha=1, beta=0 is plaid() itself;
 * (alpha, beta) = (1/nrow(X), -0.5) applied to raw ranks is replaid.sing (R/plaid.R:216),
 * (1/max(rX), -0.5) is replaid.ssgsea (R/plaid.R:251) -- by linearity of the crossprod.
 * X: g x n column-major, leading dimension ldx; S: m x n column-major, leading dim lds.
 * `alpha_div` (device double*, may be NULL): alpha is divided by *alpha_div on the device, so
 * the global max(rX) never visits the host.  `flags` (device uint32[4], may be NULL): see above. */
int plaidhip_dev_spmm_dense_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* X,
                                int64_t ldx, int32_t n, int stat, double alpha, const void* alpha_div,
                                double beta, void* S, int64_t lds, void* flags);
/* The same crossprod for an X that holds RANKS -- exactly what plaidhip_dev_colranks_dense_f64 writes with power = 1 and
 * is_signed = 0 (half-integers in [0.5, nrow(X)]), or such ranks after replaid.ucell's max - rank / pmin map: the rank
 * matrix of replaid.sing (R/plaid.R:215-217), replaid.ssgsea(alpha = 0) (:245-253), replaid.ucell (:277-279).  2 * rank
 * is staged as u16 (four sample columns per 8-byte LDS entry) and summed in integers: exact, order-independent and
 * bit-identical to plaidhip_dev_spmm_dense_f64 on the same input, at a quarter of its LDS bytes per score.  The launch is
 * speculative: a value that is not such a rank (NaN -- matrixStats::colRanks keeps NA --, +-Inf, a negative value, a double
 * >= 32,768) is seen while staging, and the fp64 kernel enqueued right behind it on the same stream then recomputes the
 * scores (it returns at once otherwise), so the result is plaidhip_dev_spmm_dense_f64's for ANY input, NaN propagation
 * included.  Shapes the u16 kernel does not take (nrow(X) <= 8,192 or > 20,448) run the general kernels.             */
int plaidhip_dev_spmm_ranks_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* R,
                                int64_t ldr, int32_t n, int stat, double alpha, const void* alpha_div,
                                double beta, void* S, int64_t lds, void* flags);
/* same with X as CSC (dgCMatrix) -- sparse branch of Matrix::crossprod at R/plaid.R:107.
 * `nnz`: number of stored values of X when the caller knows it (Xp[n] on the host), else -1.  It picks
 * the kernel: sparse-aware scatter below 12.5 % stored values, column gather above; with -1 both are
 * enqueued and the one that does not apply returns at once (the value is then read on the device).
 * nnz is a hint only: what the kernels read is Xx[Xp[0] .. Xp[n]).
 * The scatter kernel sums in u64 fixed point -- scores that do not depend on the order in which its LDS atomics arrive,
 * bit-identical from run to run -- when a sweep over the stored values finds them all finite and >= 0 AND their dynamic
 * range small enough for every score to stay within 2^-40 (9.1e-13) relative of the exact sum: each value is rounded once
 * to a grid of 2^-(e+1) <= 2^-40 x (smallest stored value > 0), e = 63 bits minus those of the largest possible sum
 * ((largest set size) x (largest value), or -- where that is too coarse -- the largest sum of a column's values).  Anything else (a negative, NaN or infinite value, raw
 * counts next to values near 1, one huge outlier) takes fp64 atomics.  Decided on the device; both launches are enqueued. */
int plaidhip_dev_spmm_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* Xp,
                              const void* Xi, const void* Xx, int32_t n, int64_t nnz, int stat, double alpha,
                              const void* alpha_div, double beta, void* S, int64_t lds, void* flags);

/* The sparse crossprod for RANK WEIGHTS: Rx is what plaidhip_dev_colranks_csc_f64 wrote (rank^power of the stored values,
 * so 0 <= Rx <= *rmax, rmax = their maximum on the device -- the max(rX) replaid.ssgsea divides by, R/plaid.R:251: alpha is
 * divided by it as by alpha_div).  The fixed-point grid of the scatter kernel then follows *rmax -- the maximum over the
 * WHOLE matrix -- so while the first bound applies ((largest set size) x *rmax: collections whose largest set has at most
 * 1,024 genes) every shard of a sharded call rounds alike and the scores do not depend on the sharding.  Collections with a
 * larger set fall back to the bound from the largest column sum OF THE SHARD (and the fixed-point / fp64 choice looks at the
 * shard's own smallest value): there the scores of different shardings agree to 2^-40 relative, not bit for bit.  Same
 * device-side guard as plaidhip_dev_spmm_csc_f64: a stored value outside [0, *rmax], a NaN (the rank weight of a NaN
 * input) or too wide a dynamic range takes the fp64 accumulators, which propagate it as the reference does.  rmax is
 * required.                                                                                                          */
int plaidhip_dev_spmm_csc_ranks_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* Xp,
                                    const void* Xi, const void* Rx, int32_t n, int64_t nnz, int stat, double alpha,
                                    const void* rmax, double beta, void* S, int64_t lds, void* flags);

/* chunked_crossprod(x, y) = t(x) %*% y (R/plaid.R:100-123, Matrix::crossprod at :107 / :117) for a GENERAL sparse x: the
 * stored values of x (@x) may differ inside a column -- signed or weighted gene sets -- which the prepared membership
 * of plaidhip_geneset_create cannot express.  x stays in its dgCMatrix slots (device pointers Wp: m + 1, Wi / Wx: Wp[m];
 * g rows, m columns); y is g x n dense (leading dimension ldy); S: m x n, leading dimension lds.  Every stored entry of x
 * is multiplied, explicit zeros included (0 * NaN is NaN, as in Matrix::crossprod); sums are fp64 in an order that
 * differs from a sequential one (16 partial sums per column of x).  plaid() itself never needs this entry: it builds
 * the column-scaled 0/1 matrix (:73-77), the path plaidhip_dev_spmm_dense_f64 is made for.                          */
int plaidhip_dev_crossprod_weighted_f64(plaidhip_ctx* ctx, const void* Wp, const void* Wi, const void* Wx, int32_t g,
                                        int32_t m, const void* Y, int64_t ldy, int32_t n, void* S, int64_t lds);
/* same with y a dgCMatrix (device pointers Yp: n + 1, Yi / Yx); S is dense                                           */
int plaidhip_dev_crossprod_weighted_csc_f64(plaidhip_ctx* ctx, const void* Wp, const void* Wi, const void* Wx, int32_t g,
                                            int32_t m, const void* Yp, const void* Yi, const void* Yx, int32_t n,
                                            void* S, int64_t lds);

/* colranks(), dense branch: t(matrixStats::colRanks(as.matrix(X), ties.method))
 * (R/plaid.R:611-619); `is_signed` = sign(X)*rank(|X|) (R/plaid.R:612-615).  Optional fused
 * power transform rank^power (R/plaid.R:249, power = 1+alpha; pass 1.0 for none).
 * R: g x n doubles (same layout as X).  colmax (device double[n], may be NULL) receives the
 * per-column maximum of the written values (feeds max(rX), R/plaid.R:251).              */
int plaidhip_dev_colranks_dense_f64(plaidhip_ctx* ctx, const void* X, int64_t ldx, int32_t g,
                                    int32_t n, int ties, int is_signed, double power, void* R,
                                    int64_t ldr, void* colmax);
/* sparse_colranks() (R/plaid.R:631-650): ranks of the stored non-zeros of each CSC column
 * among themselves; only @x is produced, pattern unchanged (R/plaid.R:645-646).          */
/* `max_col_nnz`: an upper bound on the number of stored values of any column (it sizes the workgroups
 * and their LDS; the number of rows of X is always valid, a tight bound is faster).  The call is
 * stream-ordered like every dev entry point: nothing is read back to size the launch.               */
int plaidhip_dev_colranks_csc_f64(plaidhip_ctx* ctx, const void* Xp, const void* Xx, int32_t n,
                                  int32_t max_col_nnz, int ties, int is_signed, double power, void* Rx,
                                  void* colmax);

/* colranks() on a dgCMatrix WITHOUT keep.zero (R/plaid.R:602-609 -> sparseMatrixStats::colRanks):
 * the zeros are ranked too and the result is DENSE g x n -- same numbers as the dense branch on
 * the densified matrix, computed from the CSC arrays on the device.                        */
int plaidhip_dev_colranks_csc_dense_f64(plaidhip_ctx* ctx, const void* Xp, const void* Xi, const void* Xx,
                                        int32_t g, int32_t n, int ties, int is_signed, double power,
                                        void* R, int64_t ldr, void* colmax);

/* The same dense ranks WITHOUT densifying the column: all zeros of a sparse column tie, so its dense ranks follow from the
 * ranks among the stored values (the sparse_colranks kernel) and the counts of negative and zero entries -- O(nnz) work plus
 * one dense write, for any nrow(X) (the densify-and-rank route leaves the fast rank kernel beyond 20,352 rows; a 10x
 * Genomics matrix has 33,538 or 36,601).  max_col_nnz: the longest column's stored values (<= 20,352, else EUNSUPPORTED);
 * Rx_scratch: Xp[n] doubles of device scratch.  Results are identical to plaidhip_dev_colranks_csc_dense_f64.          */
int plaidhip_dev_colranks_csc_dense_nz_f64(plaidhip_ctx* ctx, const void* Xp, const void* Xi, const void* Xx, int32_t g,
                                           int32_t n, int32_t max_col_nnz, int ties, int is_signed, double power,
                                           void* Rx_scratch, void* R, int64_t ldr, void* colmax);

/* normalize_medians() (R/plaid.R:554-575) in three phases so that a sample-sharded host
 * can all-reduce between them:
 *   1. flags  : plaidhip_dev_minflags   (or the SpMM epilogue's `flags`)  -> ignore.zero
 *   2. medians: plaidhip_dev_col_medians  (zeros masked when ignore_zero, all-masked
 *               column -> 0, R/plaid.R:561-566).  ignore_zero = 0 / 1, or -1 to resolve
 *               min(x)==0 on the device from `flags`.  plaidhip_dev_sum -> {sum, #non-NaN}.
 *   3. shift  : x - med[col] + add  (R/plaid.R:572); with `red` (device double[2] = {sum,
 *               count}) non-NULL, add = red[0]/red[1] is taken on the device instead.
 * Nothing in the chain needs a host round trip.                                           */
int plaidhip_dev_minflags(plaidhip_ctx* ctx, const void* S, int64_t count, void* flags);
int plaidhip_dev_col_medians(plaidhip_ctx* ctx, const void* S, int64_t lds, int32_t m, int32_t n,
                             int ignore_zero, const void* flags, void* med);
/* The sparse crossprod that ALSO prepares normalize_medians (round 4).  plaidhip_dev_spmm_csc_fused_f64 is
 * plaidhip_dev_spmm_csc_f64 (rmax == NULL) or plaidhip_dev_spmm_csc_ranks_f64 (rmax != NULL) -- same S, same flags --
 * and, when the scatter kernel takes the input and the result has more than 6,144 sets per column, it classifies every
 * score it writes against a bracket around the column's median (predicted from the column's mean score, which is known from
 * X before the crossprod, and calibrated on the first 256 columns): counts below / zero / NaN and the 1-5 % of the scores
 * inside the bracket go to a scratch the context owns.  plaidhip_dev_col_medians_resume then is plaidhip_dev_col_medians
 * for that S: the medians are selected among the candidates -- the same two middle values, bit for bit -- and only columns
 * whose bracket missed (or everything, if the matrix turns out to follow the other ignore.zero rule than the calibration
 * columns) are read again by the standalone kernel.  The 40 GB second pass of config 3 is gone.  Call it after the flag
 * words are final (a sample-sharded host all-reduces them in between) and before S is changed; with any other S, or after
 * an ineligible crossprod, it simply is plaidhip_dev_col_medians.  nnz must be the caller's true count (>= 0) for the
 * fused form to apply.  Stream-ordered, no host round trip -- except that the context's candidate scratch is (re)allocated
 * when a call needs more than the last one did (hipStreamSynchronize + hipFree + hipMalloc: NOT capture-safe; make the
 * first call of a shape outside a stream capture).  Memory: 8 bytes per candidate slot, min(8,192, max(1,024, 0.16 m))
 * slots per column, + 16 bytes per (column, wavefront slice): at most 0.2 x the bytes of S, kept by the context until
 * plaidhip_dev_fused_medians_discard or plaidhip_destroy.  If that allocation fails the plain crossprod runs.             */
int plaidhip_dev_spmm_csc_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* Xp, const void* Xi,
                                    const void* Xx, int32_t n, int64_t nnz, int stat, double alpha, const void* alpha_div,
                                    double beta, void* S, int64_t lds, void* flags, const void* rmax);
int plaidhip_dev_col_medians_resume(plaidhip_ctx* ctx, const void* S, int64_t lds, int32_t m, int32_t n, int ignore_zero,
                                    const void* flags, void* med);
/* The DENSE crossprod that also prepares normalize_medians (round 5): plaidhip_dev_spmm_dense_f64 -- same S, same flags --
 * and, when the fp64 pair kernel takes the input and the result has more than 6,144 sets per column and at least 1e9
 * scores (PLAIDHIP_OPT_FUSED_MEDIANS overrides the size rule), the workgroup of a column pair computes the pair's mean
 * scores from the X it stages (no extra pass over X), and the tile ends of the last gene slice classify the scores they
 * write against the bracket around (mean + calibrated offset) exactly like the sparse form above.  Finish with
 * plaidhip_dev_col_medians_resume (or ..._resume_token); with an ineligible call it is the plain crossprod.
 * Eligible means ALL of: PLAIDHIP_OPT_FUSED_MEDIANS != 2; m > 6,144; n >= 1,024 (four times the 256 calibration columns);
 * m * n >= 1e9 or PLAIDHIP_OPT_FUSED_MEDIANS == 1; flags != NULL; the fp64 pair kernel takes the input -- not the u16 / fp32
 * stagings of rank inputs (plaidhip_dev_spmm_ranks_f64, PLAIDHIP_OPT_RANKS_F32 >= 1 with a one-slice plan) or of
 * PLAIDHIP_PRECISION_MIXED, not the MFMA backend (PLAIDHIP_OPT_SPMM_DENSE_KERNEL == 3), not the one-column kernel.
 * Scratch, its memory cost and the capture caveat: as for plaidhip_dev_spmm_csc_fused_f64 above.                        */
int plaidhip_dev_spmm_dense_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* X, int64_t ldx, int32_t n,
                                      int stat, double alpha, const void* alpha_div, double beta, void* S, int64_t lds,
                                      void* flags);
/* plaidhip_dev_col_medians_resume recognises "that S" by (pointer, lds, m, n) only: right for a caller that resumes directly
 * after the fused crossprod.  A caller that may free and re-allocate S in between (an allocator that reuses addresses)
 * takes the TOKEN of the fused launch (plaidhip_dev_fused_medians_info, info[3], right after the crossprod; 0 = the plain
 * route ran, nothing is pending) and resumes with it: the candidates are used only if they are still the pending ones of
 * exactly that launch; any other token (0, stale) runs plaidhip_dev_col_medians -- always correct -- and drops what was
 * pending.  plaidhip_dev_fused_medians_discard drops it explicitly (a caller that will not normalise after all) and
 * releases the candidate scratch (it waits for the context's stream first).                                             */
int plaidhip_dev_col_medians_resume_token(plaidhip_ctx* ctx, int64_t token, const void* S, int64_t lds, int32_t m, int32_t n,
                                          int ignore_zero, const void* flags, void* med);
int plaidhip_dev_fused_medians_discard(plaidhip_ctx* ctx);
/* what the last fused crossprod of this context left behind (tests, tools): info[0] = its number of columns (0: it ran the
 * plain route), info[1] = device pointer to int32 status[n] (after ..._resume: 1 = median selected from the candidates,
 * 0 = the standalone kernel computed it), info[2] = device pointer to the calibration {offset, half width, ignore-zero},
 * info[3] = the launch's token (> 0) while a resume is pending, else 0.                                               */
int plaidhip_dev_fused_medians_info(plaidhip_ctx* ctx, int64_t info[4]);
int plaidhip_dev_sum(plaidhip_ctx* ctx, const void* v, int64_t count, void* out /* double[2]: sum, #non-NaN */);
int plaidhip_dev_shift_columns(plaidhip_ctx* ctx, void* S, int64_t lds, int32_t m, int32_t n,
                               const void* med, double add, const void* red);
/* phase 3 fused with an fp64 -> fp32 cast into ANOTHER matrix (float out[ldo * n]): out = (float)((S - med[col]) + add), S left
 * as it is.  What a sample-sharded job sends to the root when the assembled result must be fp32 to fit one GPU (config 5:
 * 1e6 cells x 50,000 sets = 400 GB in fp64): one read of S and a half-size write instead of shift (read + write) followed by a
 * cast (read + half-size write).  Bit-identical to plaidhip_dev_shift_columns followed by a conversion to float.
 * R/plaid.R:572 (the sweep) and :110-119 (the chunks the reference assembles).                                          */
int plaidhip_dev_shift_columns_cast_f32(plaidhip_ctx* ctx, const void* S, int64_t lds, int32_t m, int32_t n, const void* med,
                                        double add, const void* red, void* out, int64_t ldo);
/* max over a device double vector (global max(rX), R/plaid.R:251) */
int plaidhip_dev_max(plaidhip_ctx* ctx, const void* v, int64_t count, void* out /* double[1] */);

/* ---- host-level entry points: what the R `.Call` shim binds (r-pkg/src/plaidhip_R.c) -- */

/* plaid(X, matG, stats, chunk=NULL, normalize) body, R/plaid.R:73-85, dense X.
 * S_out: m x n doubles, caller-allocated.                                               */
int plaidhip_plaid_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n,
                         const int32_t* Gp, const int32_t* Gi, int32_t m, int stat, int normalize,
                         double* S_out);
/* same for a dgCMatrix X */
int plaidhip_plaid_csc(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                       int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                       int stat, int normalize, double* S_out);
/* chunked_crossprod(x, y) with a general sparse x (see plaidhip_dev_crossprod_weighted_f64), host pointers: x as
 * dgCMatrix slots, y dense g x n; S_out m x n, caller-allocated.  The caller's chunk loop (R/plaid.R:110-119) bounds n. */
int plaidhip_crossprod_weighted_dense(plaidhip_ctx* ctx, const int32_t* Wp, const int32_t* Wi, const double* Wx,
                                      int32_t g, int32_t m, const double* Y, int32_t n, double* S_out);
/* same for a dgCMatrix y */
int plaidhip_crossprod_weighted_csc(plaidhip_ctx* ctx, const int32_t* Wp, const int32_t* Wi, const double* Wx,
                                    int32_t g, int32_t m, const int32_t* Yp, const int32_t* Yi, const double* Yx,
                                    int32_t n, double* S_out);
/* normalize_medians(x, ignore.zero), R/plaid.R:554-575, in place; med_out (n) may be NULL */
int plaidhip_normalize_medians(plaidhip_ctx* ctx, double* S, int32_t m, int32_t n, int ignore_zero,
                               double* med_out);
/* colranks(X, signed, ties.method), dense branch R/plaid.R:611-619 */
int plaidhip_colranks_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n, int ties,
                            int is_signed, double* R_out);
/* sparse_colranks(X, signed, ties.method), R/plaid.R:631-650: Rx_out has Xp[n] entries */
int plaidhip_colranks_csc(plaidhip_ctx* ctx, const int32_t* Xp, const double* Xx, int32_t n,
                          int ties, int is_signed, double* Rx_out);
/* colranks(X sparse, keep.zero=FALSE), R/plaid.R:602-609: dense g x n result from CSC input */
int plaidhip_colranks_csc_dense(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                                int32_t g, int32_t n, int ties, int is_signed, double* R_out);
/* replaid.sing body, R/plaid.R:215-217 (dense X; G aligned to X's rows as above)         */
int plaidhip_sing_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n,
                        const int32_t* Gp, const int32_t* Gi, int32_t m, double* S_out);
/* replaid.sing body for a dgCMatrix X (zeros are ranked: colranks' sparse branch without keep.zero, R/plaid.R:602-609):
 * X goes to the device as its CSC slots; the reference (and R/plaid.R:215) densify it                                */
int plaidhip_sing_csc(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t g, int32_t n,
                      const int32_t* Gp, const int32_t* Gi, int32_t m, double* S_out);
/* replaid.ssgsea body, R/plaid.R:245-253, dense X                                        */
int plaidhip_ssgsea_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n,
                          const int32_t* Gp, const int32_t* Gi, int32_t m, double alpha,
                          double* S_out);
/* replaid.ssgsea body for a dgCMatrix X (rank step = sparse_colranks, R/plaid.R:600-601)  */
int plaidhip_ssgsea_csc(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                        int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                        double alpha, double* S_out);

/* ---- several GPUs of one node from ONE host process (the R session): multi.cpp ----------------------
 * The sample columns are cut into ndev contiguous shards (plaidhip_shard_bounds); a host thread per device
 * moves its shard over its own PCIe link (pipelined through pinned staging), runs the same kernels, and the
 * three scalars that couple the samples -- max(rX) (R/plaid.R:251), min(x) == 0 (:556-557), mean(medx) (:572)
 * -- are combined on the host between the phases.  Nothing else crosses between devices (no RCCL).
 * `devices`: ndev distinct device ordinals, or NULL for 0 .. ndev-1.  X: dense g x n (Xp == NULL, X_or_x are
 * the doubles) or a dgCMatrix (Xp, Xi, X_or_x = @x).  The contexts are created on first use and kept by the
 * library until plaidhip_multi_finalize().  For dense X the results equal the single-device entry points bit
 * for bit (shards only change which device computes a column).  For a dgCMatrix every sharding takes the same
 * kernel (chosen from the density of the whole matrix); the sparse-aware scatter kernel adds in arrival order, so
 * its scores agree to the last bits only (~1e-16 relative), between shardings as between two runs.             */
int plaidhip_shard_bounds(int64_t n, int ndev, int k, int64_t* lo, int64_t* hi);   /* columns [lo, hi) of shard k */
int plaidhip_plaid_multi(const int* devices, int ndev, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                         int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, int stat,
                         int normalize, double* S_out);
int plaidhip_sing_multi(const int* devices, int ndev, const double* X, int32_t g, int32_t n, const int32_t* Gp,
                        const int32_t* Gi, int32_t m, double* S_out);
/* replaid.sing for a dgCMatrix X over several devices (see plaidhip_sing_csc) */
int plaidhip_sing_csc_multi(const int* devices, int ndev, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t g,
                            int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, double* S_out);
int plaidhip_ssgsea_multi(const int* devices, int ndev, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                          int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, double alpha,
                          double* S_out);
/* precision of the dense crossprod on the library-owned contexts of the *_multi entry points (plaidhip_set_precision's
 * counterpart; default PLAIDHIP_PRECISION_F64) */
int plaidhip_multi_set_precision(int mode);
int plaidhip_multi_finalize(void);

/* ---- "next" rows of the scope table: thin callers of the same two kernels --------------- */

/* replaid.ucell(X, matG, rmax), R/plaid.R:276-282: rX = colranks(X, "average");
 * rX = pmin(max(rX) - rX, rmax + 1); S = plaid(rX, matG); S = 1 - S/rmax + (k_full + 1)/(2 rmax).
 * X dense (Xp == NULL: X_or_x is g x n doubles) or dgCMatrix (zeros are ranked, R/plaid.R:602-609).
 * k_full[m] = colSums(matG != 0) of the UN-aligned matrix, exactly as R/plaid.R:280 uses it.   */
int plaidhip_ucell(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                   int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                   const double* k_full, double rmax, double* S_out);
/* replaid.aucell(X, matG, aucMaxRank), R/plaid.R:304-309:
 * ww = 1.08 * pmax((rX - (max(rX) - K)) / K, 0); plaid(ww, matG, stats = "mean").             */
int plaidhip_aucell(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                    int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                    double auc_max_rank, double* S_out);
/* replaid.scse(X, matG, removeLog2, scoreMean), R/plaid.R:155-190.  remove_log2: -1 = NULL (auto:
 * min(X) == 0 && max(X) < 20, :160-161), 0, 1.  score_mean: 0 -> sum statistic, x100 (:180-182);
 * 1 -> mean statistic divided by colMeans(|X|) (:175-177).  removed_log2 (may be NULL): set to 1
 * when the 2**x transform ran -- the automatic decision is taken on the device; the host prints
 * the reference's message from it (:164).                                                       */
int plaidhip_scse(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                  int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                  int remove_log2, int score_mean, double* S_out, int* removed_log2);

/* replaid.gsva(X, matG, tau, rowtf), R/plaid.R:338-363, dense X: row transform (rowtf = 0: "z",
 * center + scale per gene; 1: "ecdf", the per-gene empirical CDF), signed average ranks per sample,
 * / max|rank|, sign * |.|^(1 + tau) for tau > 0, then plaid(mean, normalised).  The row transform
 * needs every sample of a gene, so this call does not shard by sample.                           */
int plaidhip_gsva(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n, const int32_t* Gp,
                  const int32_t* Gi, int32_t m, double tau, int rowtf, double* S_out);

/* plaid.test(X, y, G, gsetX, tests, metap.method), R/plaid.R:392-474, for dense X and the aligned
 * pattern G.  y: 0 / 1 per sample.  gsetX: sets x samples scores, or NULL: plaid(X, G) is computed
 * and stays on the device (:424-427).  tests: bit mask 1 = "one" (one-sample t on logFC, :476-486),
 * 2 = "two" (:488-520), 4 = "lm" (Welch per set over the scores, Rfast::ttests, :429).
 * metap_method: 0 = fisher / sumlog, 1 = stouffer / sumz (:522-537).
 * out: sets x 6, column-major: gsetFC, p.one, p.two, p.lm, p.meta, q.meta -- in the column order of
 * G (the caller sorts, :469-471); columns of tests that were not asked for are NaN.               */
int plaidhip_plaid_test(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n, const int32_t* y,
                        const int32_t* Gp, const int32_t* Gi, int32_t m, const double* gsetX, int tests,
                        int metap_method, double* out);

/* plaid.test over SAMPLE SHARDS (one process per GPU): the statistics of R/plaid.R:407-431 are row-wise sums over the
 * samples, so every shard reduces its own columns on the device and the caller adds the shards' results (an all-reduce
 * of 2 x rows doubles) before the host half runs once.  All pointers of the two _dev_ entries are device pointers;
 * stream-ordered, no synchronisation.
 *   plaidhip_dev_row_group_sums: sums[0 * rows + r] = sum of A[r, c] over the columns with y[c] == 0, sums[rows + r] over
 *     y[c] == 1 (rowMeans of :407-408 and :431 times the group size).  A: rows x n, column-major, leading dimension ld.
 *   plaidhip_dev_row_group_ssd: ssd[.] = sum of (A[r, c] - mean[group, r])^2 per group, mean: [2][rows] -- the caller
 *     passes the means of ALL shards, so the shards' results add up to the two-pass sums the one-device call computes
 *     (the group variances of Rfast::ttests, :429).
 *   plaidhip_plaid_test_finish (host only, no device): T = [2][m] per-set sums of fc and of fc^2 (crossprod of G with the
 *     two columns, :478-479), tot1 / tot2 = sums of fc and fc^2 over all g genes (:490-493), SM = [4][m] group-0 mean,
 *     group-1 mean, group-0 ssd, group-1 ssd of the score rows (NULL without the "lm" test), n0 / n1 the group sizes.
 *     Same `tests`, `metap_method` and `out` as plaidhip_plaid_test, which calls it.                                     */
int plaidhip_dev_row_group_sums(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n,
                                const int32_t* y, double* sums);
int plaidhip_dev_row_group_ssd(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n,
                               const int32_t* y, const double* mean, double* ssd);
int plaidhip_plaid_test_finish(int32_t g, int32_t m, const int32_t* Gp, const double* T, double tot1, double tot2,
                               const double* SM, int64_t n0, int64_t n1, int tests, int metap_method, double* out);

/* ---- GMT text -> 0/1 membership matrix on the host (no device involved) --------------------------
 * Replaces read.gmt() R/gmt-utils.R:99-125 and gmt2mat() R/gmt-utils.R:19-66 (50.9 s for a 50k-set
 * collection in R, experiments/benchmark/benchmark-plaid.R:42).  Objects are owned by the library
 * until *_destroy; returned pointers stay valid until the next call on the same object.            */
typedef struct plaidhip_gmt plaidhip_gmt;         /* a named list of gene sets                        */
typedef struct plaidhip_gmtmat plaidhip_gmtmat;   /* genes x sets 0/1 dgCMatrix pattern + dimnames    */

/* read.gmt(gmt.file, add.source, nrows): '#' comments, tab fields name/source/genes, genes split on
 * ' ' or tab, "" / "NA" / repeats dropped.  nrows <= 0: all lines.                                 */
int plaidhip_gmt_read(const char* path, int add_source, int64_t nrows, plaidhip_gmt** out);
/* the same from memory.  raw != 0: exchange format for an in-memory list (one set per line,
 * name TAB source TAB gene TAB gene ...; nothing is filtered but empty tokens and repeats)         */
int plaidhip_gmt_parse(const char* text, int64_t nbytes, int raw, int add_source, int64_t nrows,
                       plaidhip_gmt** out);
int64_t plaidhip_gmt_nsets(const plaidhip_gmt* gmt);
const char* plaidhip_gmt_set_name(const plaidhip_gmt* gmt, int64_t j);
int64_t plaidhip_gmt_set_size(const plaidhip_gmt* gmt, int64_t j);
const char* plaidhip_gmt_set_gene(const plaidhip_gmt* gmt, int64_t j, int64_t k);
/* all sets as text, one line per set: name TAB gene TAB gene ... (bulk transfer to a host language) */
const char* plaidhip_gmt_text(plaidhip_gmt* gmt, int64_t* nbytes);
int plaidhip_gmt_destroy(plaidhip_gmt* gmt);

/* gmt2mat(gmt, max.genes, ntop, bg): sets by decreasing size, repeated names dropped, head(ntop);
 * rows = bg (nbg names) or, nbg == 0, the genes by decreasing count (ties in name order);
 * head(max.genes) (max_genes < 0: all); rows finally by decreasing number of sets (stable).        */
int plaidhip_gmt2mat(const plaidhip_gmt* gmt, int64_t max_genes, int64_t ntop, const char* const* bg,
                     int64_t nbg, plaidhip_gmtmat** out);
int plaidhip_gmtmat_dims(const plaidhip_gmtmat* mat, int64_t dims[3]);   /* genes, sets, memberships */
const int32_t* plaidhip_gmtmat_p(const plaidhip_gmtmat* mat);             /* @p, sets + 1              */
const int32_t* plaidhip_gmtmat_i(const plaidhip_gmtmat* mat);             /* @i, sorted rows per set   */
/* newline-joined dimnames: axis 0 = genes (rows), 1 = sets (columns)                                */
const char* plaidhip_gmtmat_names(plaidhip_gmtmat* mat, int axis, int64_t* nbytes);
int plaidhip_gmtmat_destroy(plaidhip_gmtmat* mat);

#ifdef __cplusplus
}
#endif
#endif /* PLAIDHIP_H */
