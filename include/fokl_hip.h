/*
 * fokl_hip.h -- C ABI of libfokl_hip.so, the MI355X (gfx950) implementation of FoKL-GPy's
 * forward-variable-selection hot path (FoKLRoutines.FoKL.fit: basis build -> Gram / X'y -> Gibbs -> BIC).
 *
 * The reference is pure Python and has no FFI seam of its own: the hot path is closure code inside
 * FoKL.fit (/root/reference/src/FoKL/FoKLRoutines.py:1350-1760, "FR" below).  This header is the seam a
 * maintainer binds with ctypes directly under that method (INTEGRATION.md shows the stub); each entry
 * point names the reference lines whose work it replaces.  Plain pointers and sizes only, no torch types.
 *
 * Conventions
 *   - every function returns 0 (FOKL_OK) or a negative FOKL_ERR_* code; fokl_last_error() gives the text;
 *   - a context owns one HIP stream on one device; use one context per host thread;
 *   - the design matrix lives on the device as column "slots" (one basis column of N fp64 values each).
 *     Slot 0 is the intercept column of ones and slot 1 holds the observations y; both are filled by
 *     fokl_upload().  All other slots are handed out by the caller (the host driver keeps the free list);
 *   - matrices crossing the ABI are dense row-major fp64 in host memory.
 */
#ifndef FOKL_HIP_H
#define FOKL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOKL_OK             0
#define FOKL_ERR_HIP       -1   /* a HIP runtime call failed (no device, out of memory, launch failure ...) */
#define FOKL_ERR_ARG       -2   /* invalid argument (null pointer, slot out of range, size mismatch ...)      */
#define FOKL_ERR_STATE     -3   /* call out of order (e.g. build before upload)                               */
#define FOKL_ERR_COMM      -4   /* RCCL failure                                                               */
#define FOKL_ERR_NUMERIC   -5   /* sampler received non-finite spectrum / hyper-parameters                    */

#define FOKL_KERNEL_SPLINES   0 /* 'Cubic Splines'          (FR:199, FR:834-836) */
#define FOKL_KERNEL_BERNOULLI 1 /* 'Bernoulli Polynomials'  (FR:199, FR:841-843) */

#define FOKL_SLOT_ONES 0
#define FOKL_SLOT_Y    1
#define FOKL_SLOT_FIRST_FREE 2

/* ids for fokl_timing_get() */
#define FOKL_K_BASIS   0        /* K1 basis-build kernel                */
#define FOKL_K_GRAM    1        /* K2 Gram kernels, HBM-bound launches  */
#define FOKL_K_RESID   2        /* K3 residual / BIC kernel             */
#define FOKL_K_PREDICT 3        /* evaluate(): X * beta^T + order stats */
#define FOKL_K_RESID_MF 4       /* K3 without the stored columns (fokl_bic_resid_terms_launch) */
#define FOKL_K_GRAM_MFMA 5      /* K2 launches bound by the fp64 MFMA roof (2 N nr nc / peak flops > 8 N distinct columns /
                                   peak bytes, i.e. nr nc / distinct > ~39); FOKL_K_GRAM then holds the HBM-bound ones */
#define FOKL_K_GRAM_REDUCE 6    /* K2's second kernel: fixed-order sum of the per-workgroup partial blocks (bytes = slabs read) */
#define FOKL_K_COUNT   7

typedef struct fokl_ctx fokl_ctx;

/* ------------------------------------------------------------------------------------------------------ */
/* library / context                                                                                      */
/* ------------------------------------------------------------------------------------------------------ */

int fokl_version(void);
/* Number of visible HIP devices (0 and FOKL_ERR_HIP when the runtime finds none). */
int fokl_device_count(int *count);
int fokl_ctx_create(int device, fokl_ctx **out);
void fokl_ctx_destroy(fokl_ctx *ctx);
/* Text of the last error raised on `ctx` (or, with ctx == NULL, by fokl_ctx_create / pure host functions). */
const char *fokl_last_error(const fokl_ctx *ctx);
int fokl_sync(fokl_ctx *ctx);

/* ------------------------------------------------------------------------------------------------------ */
/* dataset: replaces the per-fit constants of FR:1357-1361 (_inputs_to_phind, FR:544-592) and FR:1374     */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * Copy one (shard of a) normalised dataset to the device and lay it out for the kernels:
 *   x      [n, m] row-major, already normalised to [0, 1] (what FoKL.clean produces, FR:441-507);
 *   y      [n];
 *   kernel FOKL_KERNEL_*;
 *   phis   dense coefficient table: splines [n_basis, 4, width] (width = 499 pieces, coefficient k of
 *          piece p of basis i at (i*4 + k)*width + p); Bernoulli [n_basis, width] zero padded, basis i
 *          (order i + 1) uses its first i + 2 entries (the reference's tuple-of-lists, GK:221-267, GK:308-326).
 * Inputs are transposed to structure-of-arrays on the device; the spline piece index / local coordinate
 * (FR:570-589) are recomputed inside the basis kernel from x instead of being stored.
 * Invalidates every slot; slots 0 (ones) and 1 (y) are rebuilt.
 */
int fokl_upload(fokl_ctx *ctx, const double *x, const double *y, int64_t n, int m, int kernel,
                const double *phis, int n_basis, int width);

/*
 * FoKL.clean's normalisation (FR:395, 436-437) on the device, for fits that start from the RAW dataset: fokl_stage_inputs
 * copies x [n, m] (row-major, raw) to the device and returns every column's minimum and maximum (exact; NaN-propagating
 * like np.min / np.max) -- the host derives `lows` and `spans` from them (minmax / pillow keywords, a few scalars) --
 * and fokl_upload_staged does what fokl_upload does from that copy, every value normalised on the way:
 * (x - lows[k]) / spans[k], one subtraction and one division, separately rounded: the numbers FoKL.clean produces on the
 * host bit for bit.  fokl_download_inputs returns the inputs as the kernels see them, in rows [n, m] (the `inputs`
 * attribute of the model is materialised from it when somebody asks for it).
 */
int fokl_stage_inputs(fokl_ctx *ctx, const double *x, int64_t n, int m, double *lows_out, double *highs_out);
int fokl_upload_staged(fokl_ctx *ctx, const double *y, int64_t n, int m, int kernel, const double *phis, int n_basis,
                       int width, const double *lows, const double *spans);
int fokl_download_inputs(fokl_ctx *ctx, double *x_out);

/* Make sure slots [0, n_slots) exist (grows in chunks; existing slot contents are preserved). */
int fokl_reserve_slots(fokl_ctx *ctx, int n_slots);
int fokl_slot_capacity(const fokl_ctx *ctx);
int64_t fokl_rows(const fokl_ctx *ctx);

/*
 * FoKL.clean's two passes over a row-major dataset x [n, m] on `threads` host threads (FR:395, 436-437): the minimum and
 * maximum of every column (exact; NaN-propagating like np.min / np.max), and x <- (x - lows) / spans in place -- one
 * subtraction and one division per element, separately rounded: the reference's numbers bit for bit.
 */
int fokl_column_min_max(const double *x, int64_t n, int m, double *lows, double *highs, int threads);
int fokl_normalize_columns(double *x, int64_t n, int m, const double *lows, const double *spans, int threads);

/* ------------------------------------------------------------------------------------------------------ */
/* K1: basis-matrix columns.  Replaces the X-build triple loop of gibbs(), FR:1446-1485, with              */
/* evaluate_basis (FR:807-849, d = 0) and _inputs_to_phind (FR:570-589) fused in.                          */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * terms [T, m] int32 row-major: entry = basis order (1-based) of that input in the term, 0 = input absent
 *       (a row of the reference's interaction matrix `discmtx`, FR:1473);
 * slots [T]: destination slot of each term's column.
 * Column j = prod_{k: terms[j,k] != 0} basis_{terms[j,k]}(x[:, k]), factors multiplied in ascending k and
 * every operation rounded separately exactly like the reference's scalar code (no FMA contraction,
 * x**k correctly rounded).  Asynchronous on the context's stream.
 */
int fokl_build_terms(fokl_ctx *ctx, const int32_t *terms, int T, const int32_t *slots);

/*
 * The same columns with the factor of input `wrt_input` replaced by its `order`-th derivative (1 or 2) divided
 * by `divisor`: replaces the per-term product of bss_derivatives (FR:764-787; evaluate_basis d = 1, 2, FR:837-847).
 * As in the reference every factor is evaluated at the twice-normalised coordinate X of FR:584-586 (splines), the
 * caller passes divisor = (span_m / l) ** order (FR:758-759) and leaves out terms that do not contain `wrt_input`
 * (their derivative is zero, FR:785-787).
 */
int fokl_build_terms_deriv(fokl_ctx *ctx, const int32_t *terms, int T, const int32_t *slots, int wrt_input,
                           int order, double divisor);

/* ------------------------------------------------------------------------------------------------------ */
/* K2: Gram blocks.  Replaces XtX = X'X, Xty = X'y (FR:1492-1494) and dtd = y'y (FR:1374).                 */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * out[a, b] = sum_i col(row_slots[a])[i] * col(col_slots[b])[i]   (nr x nc, row-major, host memory).
 * Passing FOKL_SLOT_Y among the column slots yields X'y, FOKL_SLOT_ONES yields column sums.
 * `path`: 0 = choose automatically, 1 = force the wavefront-reduction (VALU) kernel, 2 = force the fp64
 * MFMA kernels (lists of 16 x 16 tiles; where row-side columns reappear on the column side only the tiles on or
 * above the diagonal are computed and the rest is mirrored), 3 = the earlier MFMA kernel over rectangular panels
 * (kept for A/B runs).  Partial sums are combined in a fixed order, so results are bitwise reproducible.
 * If a communicator is attached (fokl_comm_init) and `allreduce` != 0 the block is summed over ranks
 * (row-sharded data) before it is returned.  Blocking.
 */
int fokl_gram(fokl_ctx *ctx, const int32_t *row_slots, int nr, const int32_t *col_slots, int nc,
              double *out, int path, int allreduce);
/*
 * The same block in two halves (path 0): the launch returns at once and other work may be launched behind it on the
 * context's stream (residual passes, basis builds); fokl_gram_fetch waits for the block only and copies its
 * `count` = nr * nc doubles to `out`.  One block can be on its way at a time (a launch drops a block nobody fetched).  The shipped driver builds the coming
 * sub-stage's Gram block this way while the kill tests of the current one go on (FR:1492-1494 for the next call of
 * gibbs()).
 */
int fokl_gram_launch(fokl_ctx *ctx, const int32_t *row_slots, int nr, const int32_t *col_slots, int nc, int allreduce);
int fokl_gram_fetch(fokl_ctx *ctx, double *out, int64_t count);
int fokl_gram_ready(fokl_ctx *ctx);      /* 1: fokl_gram_fetch will not wait, 0: the launched block is still on its way */
/*
 * The launch plan path 2 would use for such a block -- host arithmetic only, no device needed (the CPU tests replay it
 * with numpy).  kind 0: gram_tiles_kernel (v_mfma_f64_16x16x4; what fokl_gram runs), kind 1: gram_tiles4s_kernel (the
 * 4x4x4 form of the instruction, at most 4 tiles per wavefront and 8 staged column tiles; FOKL_GRAM_MFMA4=2, A/B runs).
 * info[10] = {internal columns, i-tiles, j-tiles, groups, tiles per wavefront NT, staged column tiles CT,
 * log2(sub-chunks of 32 rows per chunk), wavefronts per tile KS, chunks in flight, wavefronts per workgroup}.  With
 * cap_groups >= groups also: icols[internal columns] (internal column -> slot: the row-side columns first), perm[nc]
 * (caller's column -> internal column), staged[groups][16] (column tile staged at each local index, -1 = none) and
 * tiles[groups][4 wavefronts][10][4] = {local row-side tile (+ 256: a tile of a ragged last row tile that the LDS-DMA
 * kernel forms as an 8 x 16 half tile; only entries 0 and 1 of a list), local column-side tile, output i-tile, output j-tile}
 * with -1, -1 for padding entries.  Any of the four may be NULL.
 */
int fokl_gram_plan(const int32_t *row_slots, int nr, const int32_t *col_slots, int nc, int kind, int32_t *info,
                   int32_t *icols, int32_t *perm, int32_t *staged, int32_t *tiles, int cap_groups);

/* ------------------------------------------------------------------------------------------------------ */
/* K3: residual moments for the BIC.  Replaces siglik = var(y - X betahat), FR:1551 (and FR:1505).         */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * out[0] = sum_i r_i, out[1] = sum_i r_i^2 with r = y - sum_j betahat[j] * col(slots[j]); the caller forms
 * the population variance out[1]/n - (out[0]/n)^2 and the BIC (FR:1553-1554).  Blocking; optional all-reduce.
 */
int fokl_bic_resid(fokl_ctx *ctx, const int32_t *slots, int nc, const double *betahat, double *out,
                   int allreduce);
/*
 * The same in two halves so that the host can run the (N-independent) Gibbs chain of a candidate while the
 * device streams its residuals: _launch enqueues kernel + copy-back on the context's stream and returns,
 * _fetch waits and delivers.  At most one launch may be outstanding, and no other blocking call
 * (fokl_gram, fokl_bic_resid, fokl_predict) may be issued on the context in between.
 */
int fokl_bic_resid_launch(fokl_ctx *ctx, const int32_t *slots, int nc, const double *betahat);
int fokl_bic_resid_fetch(fokl_ctx *ctx, double *out, int allreduce);

/*
 * Matrix-free form of fokl_bic_resid_launch: the model is given by its terms (rows of the interaction matrix,
 * [n_terms, M] int32 as for fokl_build_terms; betahat[0] belongs to the intercept, betahat[1 + j] to terms[j]) and the
 * kernel forms X betahat (FR:1551: `np.matmul(X, betahat)`) from the resident inputs: 8 N (M_used + 1) bytes of HBM
 * traffic instead of 8 N (P + 2), and no column needs to exist.  The model's distinct (input, order) factors are
 * evaluated once per row with the operations of fokl_build_terms, and the fit is their quadratic form
 * c0 + sum_a f_a (l_a + sum_{b > a} Q_ab f_b) -- the same moments as the stored-column pass up to the association of
 * the sum (agreement ~1e-15 of the moments' scale, not bit for bit).  Limits: one or two inputs per term; the factors
 * must fit one of the slot layouts (inputs x orders per input) 8 x 1, 16 x 1, 8 x 2, 4 x 4, 2 x 8, 8 x 4, 16 x 2, 4 x 8
 * (FOKL_RESID_TERMS_MAX_FACTORS slots); Bernoulli orders up to FOKL_RESID_TERMS_MAX_ORDER.  FOKL_ERR_ARG beyond:
 * callers take the stored-column pass.  Fetch with fokl_bic_resid_fetch.
 */
#define FOKL_RESID_TERMS_MAX_FACTORS 32
#define FOKL_RESID_TERMS_MAX_ORDER 8
int fokl_bic_resid_terms_launch(fokl_ctx *ctx, const int32_t *terms, int n_terms, const double *betahat);

/* ------------------------------------------------------------------------------------------------------ */
/* evaluate(): posterior-mean prediction and 95 % bounds on the device.  Replaces FR:966-978.              */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * For the rows currently uploaded: modells[i, d] = sum_j betas[d, j] * col(slots[j])[i] (FR:966-968),
 * mean[i] = mean_d modells[i, d] (FR:969) and, when `bounds` != NULL, bounds[i] = (sorted[cut],
 * sorted[draws - cut]) of row i's draws (FR:973-977).  betas is [draws, nc] row-major in host memory.
 */
int fokl_predict(fokl_ctx *ctx, const int32_t *slots, int nc, const double *betas, int draws, int cut,
                 double *mean, double *bounds);

/* ------------------------------------------------------------------------------------------------------ */
/* test / debug access to slots                                                                           */
/* ------------------------------------------------------------------------------------------------------ */

int fokl_read_slot(fokl_ctx *ctx, int slot, int64_t row0, int64_t nrows, double *host);
int fokl_write_slot(fokl_ctx *ctx, int slot, int64_t row0, int64_t nrows, const double *host);

/* ------------------------------------------------------------------------------------------------------ */
/* per-kernel timing with HIP events on the context's stream (bench.py's roofline figures)                 */
/* ------------------------------------------------------------------------------------------------------ */

int fokl_timing_enable(fokl_ctx *ctx, int on);
int fokl_timing_reset(fokl_ctx *ctx);
/*
 * Accumulated device time (ms), launch count and algorithmic bytes / flops of kernel family `kernel_id`, and
 * ideal_ms = the sum over its launches of the roofline time max(bytes / HBM peak, flops / fp64 MFMA peak): a family
 * whose launches sit on both sides of the ridge (the Gram blocks) is priced launch by launch.
 */
#define FOKL_PEAK_HBM_BYTES_PER_S 8.0e12  /* MI355X HBM3E, spec */
#define FOKL_PEAK_F64_FLOPS 78.6e12       /* MI355X dense fp64 (vector = matrix), spec */
int fokl_timing_get(fokl_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches, double *bytes,
                    double *flops, double *ideal_ms);

/*
 * What the device sustains, measured with trivial kernels next to the real ones: HBM read-only / write-only streams,
 * one read stream feeding seven write streams (K1's 8-input, 56-term shape) in bytes/s, and v_mfma_f64_16x16x4_f64 on
 * register operands in flop/s.  About 0.1 s and up to 4 GiB of scratch per call; not used by any computation.
 */
#define FOKL_PROBE_HBM_READ 0
#define FOKL_PROBE_HBM_WRITE 1
#define FOKL_PROBE_HBM_MIX 2
#define FOKL_PROBE_MFMA_F64 3
int fokl_probe(fokl_ctx *ctx, int what, double *rate);

/* ------------------------------------------------------------------------------------------------------ */
/* G2/G3: the Gibbs chain (host C++, N-independent).  Replaces the D-iteration loop FR:1519-1548.           */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * Runs `draws` iterations in the eigenbasis of XtX = Q diag(lamb) Q':
 *     d   = 1 / (lamb + 1/tausqd)                                   (FR:1521-1522)
 *     w   = d * qty + sqrt(sigsqd) * sqrt(d) * vec,  vec ~ N(0, I)  (FR:1524-1528; beta = Q w)
 *     bstar = b + (w'diag(lamb)w - 2 w'qty + dtd + w'w / tausqd)/2  (FR:1532-1533)
 *     sigsqd = 1 / Gamma(astar, 1/bstar)   (NaN, and no draw, if bstar < 0; FR:1538-1541)
 *     tausqd = 1 / Gamma(atau_star, 1/(w'w/(2 sigsqd) + btau))      (FR:1545-1547)
 * with qty = Q'Xty.  Random numbers come from numpy's legacy global stream, continued bit for bit:
 * mt_key[624] / mt_pos / has_gauss / gauss_cache are the fields of np.random.get_state() and are updated
 * in place (MT19937 -> 53-bit doubles -> polar Gaussian with one cached value -> Marsaglia-Tsang gamma).
 * w_out [draws, p1] receives w per iteration (the caller forms betas = w_out Q'); sigs_out / taus_out
 * [draws] may be NULL.
 */
int fokl_gibbs_chain(const double *lamb, const double *qty, int p1, double astar, double atau_star,
                     double b, double btau, double dtd, double sigsqd0, double tausqd0, int draws,
                     uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                     double *w_out, double *sigs_out, double *taus_out);

/*
 * The same chain in two halves, so that the serial random stream can run on a worker thread ahead of the
 * data-dependent arithmetic.  fokl_noise_tape advances the stream exactly as `draws` iterations of
 * fokl_gibbs_chain would and records, per iteration, the p1 standard normals and the two standard gamma variates
 * of shapes astar / atau_star (gam_sig_out, gam_tau_out [draws]).  The expensive half of the polar method is left
 * to the consumer (which can run on any thread): row k of normals_out [draws, p1] holds, after lead_out[k] (0 or 1)
 * finished values, (p1 - lead) / 2 accepted pairs as (x2, x1) with r2 = x1^2 + x2^2 in row k of pair_r2_out
 * [draws, p1 / 2 + 1] -- the normals are sqrt(-2 log(r2) / r2) * (x2, x1) -- and, if p1 - lead is odd, one more
 * finished value.  normals_out and pair_r2_out need 16 and 8 doubles of slack after their last row (the recorder
 * stores whole vectors; what it writes past a row is overwritten when the next row is recorded).
 * fokl_gibbs_chain_from_tape completes the normals and replays the arithmetic: identical w / sigs / taus, bit for
 * bit.
 * The split is exact unless some iteration has bstar < 0, where the reference skips a gamma draw (FR:1538-1539;
 * impossible for b > 0): *bstar_negative is then set to 1 and the caller must redo the candidate with
 * fokl_gibbs_chain from the stream state it saved before the tape.
 * `progress` (may be NULL) lets the consumer follow a tape that is still being recorded on another thread: the
 * producer stores the number of complete iterations (release) after every FOKL_TAPE_BLOCK of them and at the end, or
 * -1 on failure; the consumer waits (acquire) until iteration k is there.  Both sides must be given the same int32,
 * initialised to 0 before the producer starts; keep it on a cache line the producer does not otherwise write.
 */
#define FOKL_TAPE_BLOCK 16
int fokl_noise_tape(int p1, int draws, double astar, double atau_star, uint32_t *mt_key, int32_t *mt_pos,
                    int32_t *has_gauss, double *gauss_cache, double *normals_out, double *pair_r2_out,
                    int32_t *lead_out, double *gam_sig_out, double *gam_tau_out, int32_t *progress);
int fokl_gibbs_chain_from_tape(const double *lamb, const double *qty, int p1, double b, double btau, double dtd,
                               double sigsqd0, double tausqd0, int draws, const double *normals,
                               const double *pair_r2, const int32_t *lead, const double *gam_sig,
                               const double *gam_tau, double *w_out, double *sigs_out, double *taus_out,
                               int32_t *bstar_negative, const int32_t *progress);

/*
 * The consumer side split once more, so that several threads can share the log / sqrt work of ONE tape:
 * fokl_finish_tape_blocks completes, in place, the normals of iteration blocks part, part + parts, ... (`block`
 * iterations each), waiting on `progress` as above, and stores block_done[blk] = 1 (release) after each, -1 on failure;
 * fokl_gibbs_chain_from_finished_tape runs the recursion on finished normals, waiting on block_done (NULL: the whole
 * tape is finished).  Results are those of fokl_gibbs_chain_from_tape bit for bit.
 */
int fokl_finish_tape_blocks(int p1, int draws, double *normals, const double *pair_r2, const int32_t *lead,
                            const int32_t *progress, int part, int parts, int block, int32_t *block_done);
int fokl_gibbs_chain_from_finished_tape(const double *lamb, const double *qty, int p1, double b, double btau,
                                        double dtd, double sigsqd0, double tausqd0, int draws, const double *normals,
                                        const double *gam_sig, const double *gam_tau, const int32_t *block_done,
                                        int block, double *w_out, double *sigs_out, double *taus_out,
                                        int32_t *bstar_negative);

/* Raw access to the same generator (parity tests against numpy): n standard normals / n std gammas. */
int fokl_rng_normals(uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                     int64_t n, double *out);
int fokl_rng_gammas(uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                    double shape, double scale, int64_t n, double *out);

/* ------------------------------------------------------------------------------------------------------ */
/* N4: the consumer of fitted models, GP_Integrate (reference src/FoKL/GP_Integrate.py:5-282)               */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * Fourth-order Runge-Kutta integration of dy_k/dt = model_k(inputs) for n_states cubic-spline BSS-ANOVA models.
 * Model k: betas[k] [mtx_rows[k] + 1] (constant first), mtx[k] [mtx_rows[k], mtx_cols[k]] basis orders; its input
 * vector has n_source[k] >= mtx_cols[k] entries, entry i being state source[k][i] (normalised with norms
 * [2, n_states] = minima, maxima and clamped to [0, 1]) if source[k][i] >= 0, else column -(source[k][i] + 1) of the
 * current row of forcing [n_steps, n_other] (already normalised; the same row serves the four stages of a step).
 * spline_table [n_basis, 4, width] as in fokl_upload, width must be 499 (evaluated on 498 intervals, as the
 * reference does).  y [n_states] holds the initial state and is advanced in place (the reference mutates y0 too);
 * trajectory [n_states, n_steps + 1] receives the state before the first and after every step.  Host only.
 */
int fokl_gp_integrate(int n_states, int n_other, int64_t n_steps, const double *const *betas,
                      const int32_t *const *mtx, const int32_t *mtx_rows, const int32_t *mtx_cols,
                      const int32_t *const *source, const int32_t *n_source, const double *forcing,
                      const double *norms, const double *spline_table, int n_basis, int width, double h, double *y,
                      double *trajectory);

/* ------------------------------------------------------------------------------------------------------ */
/* multi-GPU: one process per GPU, RCCL over xGMI                                                          */
/* ------------------------------------------------------------------------------------------------------ */

#define FOKL_UNIQUE_ID_BYTES 128
/* Rank 0 creates the id and hands the bytes to the other ranks through any host channel. */
int fokl_comm_unique_id(char id[FOKL_UNIQUE_ID_BYTES]);
int fokl_comm_init(fokl_ctx *ctx, const char id[FOKL_UNIQUE_ID_BYTES], int rank, int world);
int fokl_comm_destroy(fokl_ctx *ctx);
/* fokl_comm_init in two steps, for launchers that put a deadline on the (collective) initialisation: the first runs
 * ncclCommInitRank on `device` and touches no context -- it may sit on a helper thread that is abandoned when the
 * deadline passes; the second attaches the communicator to the context once every rank has reported success;
 * fokl_comm_release_detached drops one that will not be used. */
int fokl_comm_init_detached(int device, const char id[FOKL_UNIQUE_ID_BYTES], int rank, int world, void **comm_out);
int fokl_comm_adopt(fokl_ctx *ctx, void *comm, int rank, int world);
int fokl_comm_release_detached(void *comm);
/* recv[r*count .. (r+1)*count) = send of rank r: the per-candidate BIC gather of the kill-test shard. */
int fokl_comm_allgather_f64(fokl_ctx *ctx, const double *send, int count, double *recv);
/* In-place sum over ranks (row-sharded Gram blocks / residual moments). */
int fokl_comm_allreduce_sum_f64(fokl_ctx *ctx, double *buf, int count);

#ifdef __cplusplus
}
#endif
#endif /* FOKL_HIP_H */
