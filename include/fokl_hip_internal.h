/*
 * fokl_hip_internal.h -- what libfokl_hip.so exports BESIDE its C ABI (include/fokl_hip.h): the plumbing of this
 * package's own host pipeline, bound by fokl_gpy_amd/_capi.py and by nothing else.  Not part of the boundary a maintainer of
 * the reference would bind (INTEGRATION.md binds fokl_hip.h only) and free to change between releases:
 *
 *   fokl_stream_*              the numpy-legacy random stream in two phases (bulk threads + one serial walk)
 *   fokl_pool_*                the host threads of one fit (noise / chain / finish / spectral queues)
 *   fokl_search_* / fokl_outcome_* / fokl_spectrum_*
 *                              the search's per-evaluation work next to the kill-test loop (FR:1650-1690)
 *   fokl_dchain_* / fokl_dspectral_* / fokl_device_dgemm* / fokl_host_alloc
 *                              the device engines behind G3 (and the opt-in G2) and their page-locked memory
 *
 * Same conventions as fokl_hip.h (return codes, row-major fp64, FR = /root/reference/src/FoKL/FoKLRoutines.py).
 */
#ifndef FOKL_HIP_INTERNAL_H
#define FOKL_HIP_INTERNAL_H

#include <stddef.h>

#include "fokl_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------ */
/* The random stream in two phases: bulk threads + one serial walk (round 4; csrc/fokl_stream.cpp).          */
/* Call sites replaced: np.random.normal FR:1527, np.random.gamma FR:1541 / FR:1547.                         */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * The stream of fokl_noise_tape, produced by `bulk_threads` threads ahead of ONE walking thread.  The bulk threads
 * continue MT19937 from the state handed to fokl_stream_create (np.random.get_state()), temper the words and flag, for
 * every double of the stream, whether the polar attempt that STARTS there is accepted (both pairings of the doubles: a
 * gamma's uniform shifts the pairing by one).  fokl_stream_walk advances over one model evaluation's draws exactly
 * as fokl_noise_tape does -- same positions, same consumption, same cached value -- but touches only what is serial: per
 * Gibbs iteration it steps over ceil((p1 - lead) / 2) accepted attempts by counting flags and makes the two gamma draws
 * -- and those from BOUNDS where they decide (a normal is known by its source, the accepted attempt it comes from and
 * which half; its value is formed, with libm's log, only when a gamma's accept test is too close to call: 1 in 10).  A
 * tape row is 32 bytes of positions:
 *     start        double index where the iteration's attempts begin; bit 63 = the row opens with the cached normal
 *     lead_source  the attempt whose x1 half that cached normal is
 *     gamma[2]     source of the normal X the accepted attempt of each gamma draw used (bit 63 = the x1 half): the
 *                  variate is b (1 + c X)^3; all ones = the walker stored the variate itself (shapes <= 1)
 * fokl_stream_expand turns rows back into fokl_noise_tape's layout (raw pairs (x2, x1), r2, lead, final lead / tail
 * values, the two gamma variates) on any thread -- values identical to fokl_noise_tape's; the part of the stream a tape
 * covers must be HELD from before its walk until its last expansion: fokl_stream_hold (on the walking thread, at the
 * position the tape starts from) / fokl_stream_release (any thread).  The walker itself keeps everything from its floor
 * on; fokl_stream_advance_floor moves the floor to its present position (between tapes).
 * fokl_stream_tell / fokl_stream_seek save and restore the walker (tentative tapes: a rewind is three words, the
 * bulk data does not move).  fokl_stream_state writes numpy's state tuple at the walker's position.
 * One thread at a time may walk / seek / tell / ask for the state; expand, hold-release and stats are thread-safe.
 */
#define FOKL_SEGMENT_BLOCKS 256                      /* MT19937 blocks per segment of the stream */
#define FOKL_SEGMENT_DOUBLES 79872                  /* = 256 * 624 / 2 doubles per segment */
#define FOKL_ROW_LEAD (1ull << 63)                  /* fokl_tape_row.start: the row opens with the cached normal */
#define FOKL_SOURCE_X1_HALF (1ull << 63)            /* a normal's source: the x1 half of the attempt (else x2) */
#define FOKL_SOURCE_GIVEN ((1ull << 62) - 1)        /* source position: the cached normal of the state handed over */
#define FOKL_GAMMA_FINAL_VALUE (~0ull)              /* fokl_tape_row.gamma[j]: the walker stored the variate itself */
#define FOKL_PRESTATE_WORDS 640                     /* one entry of the pre-state ring (see fokl_stream_create) */
#define FOKL_PRESTATE_BLOCKS 32                     /* a pre-state is left every 32 blocks: 8 per segment */
typedef struct fokl_stream fokl_stream;
typedef struct fokl_tape_row {
    uint64_t start;
    uint64_t lead_source;
    uint64_t gamma[2];
} fokl_tape_row;
typedef struct fokl_stream_cursor {
    uint64_t position;
    uint64_t gauss_source;
    int32_t has_gauss;
} fokl_stream_cursor;
/* prestate_ring (may be NULL) [prestate_entries * FOKL_PRESTATE_WORDS]: for a second consumer that regenerates the stream
 * itself (the device: fokl_dchain_*), the bulk threads leave there, for every run of FOKL_PRESTATE_BLOCKS blocks and in
 * order, entry index % prestate_entries (index = segment * 8 + run) = the 624 raw words of the MT19937 block in front of
 * the run (the very first: block 0 itself, word 626 = 1), the index (words 624, 625) and the word parity the doubles pair
 * up from (word 627); fokl_stream_prestates_published counts the SEGMENTS whose eight entries are there. */
int fokl_stream_create(const uint32_t *mt_key, int32_t mt_pos, int32_t has_gauss, double gauss_cache, int bulk_threads,
                       uint32_t *prestate_ring, int prestate_entries, fokl_stream **out);
int64_t fokl_stream_prestates_published(const fokl_stream *stream);
double fokl_stream_given_gauss(const fokl_stream *stream);
void fokl_stream_destroy(fokl_stream *stream);
int fokl_stream_walk(fokl_stream *stream, int p1, int draws, double astar, double atau_star, fokl_tape_row *rows,
                     double *gam_sig, double *gam_tau, int32_t *progress);
int fokl_stream_tell(const fokl_stream *stream, fokl_stream_cursor *out);
int fokl_stream_seek(fokl_stream *stream, const fokl_stream_cursor *at);
int fokl_stream_hold(fokl_stream *stream, uint64_t *position_out);
int fokl_stream_release(fokl_stream *stream, uint64_t position);
int fokl_stream_advance_floor(fokl_stream *stream);
int fokl_stream_state(fokl_stream *stream, uint32_t *key_out, int32_t *pos_out, int32_t *has_gauss_out,
                      double *gauss_out);
int fokl_stream_expand(fokl_stream *stream, int p1, double astar, double atau_star, const fokl_tape_row *rows, int k0,
                       int k1, double *normals_out, double *pair_r2_out, int32_t *lead_out, double *gam_sig_out,
                       double *gam_tau_out);
/* seconds the bulk threads worked, seconds the walker waited for them, segments (79 872 doubles each) produced, gamma
 * attempts walked and how many of them needed the exact expressions */
/* Helper threads for the walk: the walking thread then only counts accepted attempts (the chase) and hands blocks of 128
 * iterations out; helpers turn ranks into positions / rows and run the gamma draws' accept tests.  count 0: none (default),
 * at most 4; cpus (may be NULL): the logical CPU of each helper (< 0: not pinned).  Call before the first walk. */
int fokl_stream_set_helpers(fokl_stream *stream, int count, const int32_t *cpus);
/* bulk thread i runs on logical CPU cpus[i % count] only (csrc/fokl_stream.cpp) */
int fokl_stream_place_bulk(fokl_stream *stream, const int32_t *cpus, int count);
int fokl_stream_stats(const fokl_stream *stream, double *bulk_busy_s, double *walker_wait_s, int64_t *segments,
                      int64_t *gamma_attempts, int64_t *gamma_attempts_exact);
/* max |fast_ln(y) - log(y)| over a sweep of (0, 1): the approximation the walker's bounds are built on (tests) */
double fokl_stream_fast_ln_error(int64_t n);

/* ------------------------------------------------------------------------------------------------------ */
/* Host threads of one fit: the work of G2/G3 that must not sit on the Python driver thread.                */
/* ------------------------------------------------------------------------------------------------------ */

typedef struct fokl_host_pool fokl_host_pool;
typedef struct fokl_host_job fokl_host_job;

/*
 * One noise thread that walks the random stream (fokl_stream_*: created here from mt_key / mt_pos / has_gauss / gauss_cache
 * -- caller storage, read now and WRITTEN BACK by fokl_pool_destroy with numpy's state after everything that was walked --
 * and produced by `bulk_threads` threads of its own) and records tapes strictly in submission order, `finish_threads`
 * threads that materialise each tape (all of them on every tape; 0 = the noise thread does it, block by block),
 * `chain_threads` threads that run the chain recursions and `spectral_threads` threads that diagonalise XtX sub-blocks.
 * `dsyevr` is the address of LAPACK's dsyevr with the Fortran calling convention and 32-bit integers (the Python side
 * passes scipy's own: scipy.linalg.cython_lapack.__pyx_capi__['dsyevr']), so that eigenpairs are those of the reference's
 * scipy.linalg.eigh call (FR:1499) bit for bit; NULL is allowed with spectral_threads == 0.
 * The threads inherit the CPU affinity of the caller, except that the noise thread is pinned to logical CPU
 * `noise_cpu` if that is >= 0 (the caller then keeps its other threads off that core).  Every buffer handed to a submit
 * call must stay alive until fokl_pool_wait has returned for that job.  fokl_pool_destroy first runs everything still queued.
 */
int fokl_pool_create(int chain_threads, int finish_threads, int spectral_threads, int bulk_threads, int noise_cpu,
                     void *dsyevr, uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                     uint32_t *prestate_ring, int prestate_entries, fokl_host_pool **out);
/* Models of from_columns columns or more are diagonalised by LAPACK's divide-and-conquer driver dsyevd (fn: its address,
 * Fortran ABI with 32-bit integers, e.g. scipy.linalg.cython_lapack's) instead of dsyevr: the same tridiagonal reduction,
 * eigenpairs within ~3e-12 of dsyevr's in the chain's noise map, 1.3-1.5 x faster from 80 columns on, 2 x at 585.  NULL or
 * 0: dsyevr (what scipy.linalg.eigh calls, FR:1499) for every size.  Call before the first spectral job. */
int fokl_pool_use_dsyevd(fokl_host_pool *pool, void *fn, int from_columns);
/* BLAS dgemm (fn: its address, Fortran ABI with 32-bit integers, e.g. scipy.linalg.cython_blas's) for the product of
 * fokl_pool_submit_spectral_update; NULL: those jobs decompose afresh.  Call before the first spectral job. */
int fokl_pool_use_dgemm(fokl_host_pool *pool, void *fn);
/* The same product on the device (csrc/fokl_dgemm_device.inc): fokl_device_dgemm has BLAS dgemm's signature and runs
 * C = A B ('N', 'N', alpha 1, beta 0; N and K of at least `from`, M of at least 16) as fp64 MFMA tiles on `device` through
 * page-locked staging buffers of the calling thread; every other call, and any device failure, goes to host_dgemm.
 * fokl_device_dgemm_configure returns its address through *entry -- what to give fokl_pool_use_dgemm for searches whose
 * models have hundreds of columns (585: 0.4 GFLOP per derived model, 8-10 ms on a host core).  One configuration per
 * process.  Replaces the call of scipy's dgemm inside the eigen-update (no reference line: the reference decomposes every
 * model afresh, FR:1499). */
void fokl_device_dgemm(char *transa, char *transb, int *m, int *n, int *k, double *alpha, double *a, int *lda, double *b,
                       int *ldb, double *beta, double *c, int *ldc);
int fokl_device_dgemm_configure(int device, void *host_dgemm, int from, void **entry);
int fokl_device_dgemm_stats(int64_t *calls, int64_t *on_device, int64_t *failed);

/* CPUs for the spectral threads alone (they share no data with the threads around the random stream: another last-level
 * cache domain keeps them off those threads' cores) */
int fokl_pool_spectral_affinity(fokl_host_pool *pool, const int32_t *cpus, int count);
/* the pool's stream (fokl_stream_expand of rows-only tapes; alive as long as the pool) */
fokl_stream *fokl_pool_stream(fokl_host_pool *pool);
void fokl_pool_destroy(fokl_host_pool *pool);
/* 1 = every row of the tape is recorded (progress == draws) and every block of `block` rows is there (block_done[] != 0:
 * expanded / finished by the pool's finish threads), 0 = not yet, -1 = it never will be.  Acquire loads: a consumer on any
 * thread may poll this before it reads the tape's arrays.  Either pointer may be NULL (not looked at). */
int fokl_tape_ready(const int32_t *progress, int draws, const int32_t *block_done, int block);

/*
 * One model evaluation's tape on the noise thread: fokl_stream_walk into rows [draws] (progress must be given and start
 * at 0: rows walked so far), materialised into fokl_noise_tape's layout (normals / pair_r2 / lead / gam_sig / gam_tau:
 * identical numbers) by the finish threads, block by block of `block` rows behind the walk: block_done [ceil(draws /
 * block)] (zero-initialised) receives 1 (release) per block, -1 if the tape is sent back.  finish != 0: the normals of
 * each block are also completed IN PLACE (the log / sqrt half of the polar method: tapes a host chain reads); the job
 * counts as run only when the finish threads have left the tape too.  finish == 2: ROWS ONLY -- nobody materialises
 * the tape here (normals / pair_r2 / lead / block_done may be NULL): a consumer that has the stream itself expands the
 * rows (the device: fokl_dchain_submit_rows; or fokl_stream_expand on fokl_pool_stream).
 * span_out (may be NULL; required with finish == 2) receives [position the stream is held from for this tape, walker
 * position behind the tape] before progress reaches `draws`; the hold is then the CALLER's: fokl_pool_release_hold
 * when nobody will expand the rows any more.
 * tentative != 0: the tape is walked ahead of the decision that it is needed, and fokl_pool_resolve(job, commit) is
 * its verdict -- commit keeps the tape (identical to a plain submission at that point of the stream), otherwise the
 * walker is put back where the tape began and `progress` is set to -1.  Tentative tapes may be NESTED: up to 16 can be
 * on record without a verdict, the noise thread goes on walking behind them.  A commit of the oldest makes it final;
 * an abort takes every younger tentative tape with it (the caller resolves those to "abort" as well: what they hold is
 * no longer what the stream serves there); an abort of the youngest rewinds just that one.  A plain request waits until
 * nothing tentative is left.  Every tentative job MUST be resolved, or the noise thread (and fokl_pool_destroy) waits
 * for ever.
 */
int fokl_pool_submit_noise(fokl_host_pool *pool, int p1, int draws, double astar, double atau_star, fokl_tape_row *rows,
                           double *normals, double *pair_r2, int32_t *lead, double *gam_sig, double *gam_tau,
                           int32_t *progress, int tentative, int32_t *block_done, int block, int finish,
                           uint64_t *span_out, fokl_host_job **out);
int fokl_pool_release_hold(fokl_host_pool *pool, uint64_t position);
int fokl_pool_resolve(fokl_host_job *job, int commit);
/*
 * The draws of one candidate from its tape (whose noise job must have been submitted with the same block_done): the
 * recursion follows the flags of the finish threads -- on normals they completed in place (finishing_requested != 0:
 * the tape was submitted with finish != 0) or completing each row itself.  then (may be NULL): called with then_arg by the
 * chain thread once the chain has run without error, before the job counts as done (what follows from the complete draws).
 */
int fokl_pool_submit_chain(fokl_host_pool *pool, const double *lamb, const double *qty, int p1, double b, double btau,
                           double dtd, double sigsqd0, double tausqd0, int draws, const double *normals,
                           const double *pair_r2, const int32_t *lead, const double *gam_sig, const double *gam_tau,
                           const int32_t *progress, int32_t *block_done, int block, int finishing_requested,
                           double *w_out, int32_t *bstar_negative, void (*then)(void *), void *then_arg, fokl_host_job **out);
/*
 * G2 for the candidate model made of columns idx[0..p1) of `gram` (row-major, leading dimension ld, y in column
 * ycol): XtX = gram[idx][:, idx], Xty = gram[idx, ycol].  Outputs: lamb_out [p1] ascending eigenvalues,
 * qt_out [p1, p1] with ROW j = eigenvector j (largest-magnitude component positive), qty_out = Q'Xty,
 * betahat_out = Q (qty / lamb) (FR:1499-1504).  moments_out [2] (may be NULL) receives sum r and sum r^2 of the
 * residual r = y - X betahat, formed from the Gram alone -- sum y - 1'X b and y'y - 2 b'Xty + b'XtX b in extended
 * precision; column 0 of gram must be the ones column -- the quantities fokl_bic_resid measures on the device
 * (FR:1551).  No random numbers: may be submitted speculatively.
 */
int fokl_pool_submit_spectral(fokl_host_pool *pool, const double *gram, int ld, const int32_t *idx, int p1, int ycol,
                              double *lamb_out, double *qt_out, double *qty_out, double *betahat_out,
                              double *moments_out, fokl_host_job **out);
/* G2 of a kill test's model (FR:1666-1690 evaluate the current model minus one term) from the eigenpairs of the model with
 * that ONE MORE column instead of from scratch: parent_lamb [p1 + 1] ascending, parent_qt [p1 + 1, p1 + 1] as qt_out above,
 * parent_pos = which of the parent's columns this model lacks.  The eigenvalues are the roots of the secular equation
 * sum_j z_j^2 / (lam_j - mu) = 0 (z = row parent_pos of Q), the eigenvectors one (p1 x (p1+1) x p1) dgemm; vectors from the
 * z^ of Gu & Eisenstat, so orthogonal to working precision.  parent_job: NULL when the parent's arrays are complete, else
 * the spectral job of this pool that writes them (not waited for yet): this job is queued, ahead of everything else, when
 * that one has run.  updated (may be NULL): 1 = derived from the parent, 0 = decomposed afresh after all (the parent
 * failed, nearly repeated eigenvalues or vanishing z_j -- no deflation here --, a failed check of diag(XtX) against the
 * eigenpairs, no dgemm bound, FOKL_EIGH_SIGNS=lapack).  Same outputs, layout and sign convention as
 * fokl_pool_submit_spectral; accuracy: tests/stress/eigen_deletion_study.py. */
int fokl_pool_submit_spectral_update(fokl_host_pool *pool, const double *gram, int ld, const int32_t *idx, int p1, int ycol,
                                     const double *parent_lamb, const double *parent_qt, int parent_pos,
                                     fokl_host_job *parent_job, double *lamb_out, double *qt_out, double *qty_out,
                                     double *betahat_out, double *moments_out, int32_t *updated, fokl_host_job **out);
/* 1 if the job has run.  fokl_pool_wait blocks until then, frees the job and returns its status. */
int fokl_pool_poll(const fokl_host_job *job);
int fokl_pool_wait(fokl_host_job *job);
/* Accumulated time (s) the kinds of thread spent inside jobs (including their waits on the tape producer). */
int fokl_pool_busy_seconds(const fokl_host_pool *pool, double *noise, double *chain, double *finish,
                           double *spectral);
/* The pool's random stream: fokl_stream_stats of it. */
int fokl_pool_stream_stats(const fokl_host_pool *pool, double *bulk_busy_s, double *walker_wait_s, int64_t *segments,
                           int64_t *gamma_attempts, int64_t *gamma_attempts_exact);
/* Seconds the noise thread spent waiting: with an empty queue, and for the verdict on tentative tapes. */
int fokl_pool_noise_waits(const fokl_host_pool *pool, double *queue_wait, double *verdict_wait);
/* CPU-seconds the library's own threads have used since it was loaded, by kind, process-wide (a pool's threads are added
 * when they end, i.e. when their fit's pool is destroyed): seconds[0..5] = the stream's walker, chain threads, finish
 * threads, spectral threads, the stream's bulk threads, the device-chain dispatchers (live).  -> 6, or an error.  What a fit
 * costs in CPU and where: the figure that bounds fits running side by side on a host with a CPU quota. */
int fokl_thread_cpu_seconds(double *seconds, int count);

/* ------------------------------------------------------------------------------------------------------ */
/* The search's per-evaluation work off the driver thread: tapes on order, G2 ahead, chains, kill tests     */
/* (csrc/fokl_search.cpp).  Replaces the loop FR:1666-1690 and the bookkeeping around FR:1650 / FR:1681.     */
/* ------------------------------------------------------------------------------------------------------ */

/*
 * A fokl_search holds, for ONE fit, what the sequential decisions of the forward selection share: the queue of noise
 * tapes on order ahead of the decisions that they are needed, the G2 jobs submitted ahead, the chains (pool threads or
 * the device engine), the decisions taken from guessed intercept scales and their confirmation, the BIC cache of models
 * scored before, the trace of evaluations.  It is driven by one thread.  The driver (Python: engine.ForwardSelection)
 * keeps the sequence of sub-stages, the K1 / K2 / K3 launches, the statistics that order a sub-stage's proposals and the
 * stop rule, and calls
 *   fokl_search_set_substage        term ids of the active columns (identical models score identically, see
 *                                   fokl_search_score);
 *   fokl_search_model_begin/_commit a sub-stage's model (FR:1650) around the driver's residual pass;
 *   fokl_search_score               BIC (FR:1551-1554, 1653-1654) from residual moments + the trace record;
 *   fokl_search_kill_tests          FR:1666-1690 for one sub-stage: same tests, same order, same consumption of the
 *                                   random stream; BIC of the candidates from the sub-stage's Gram.
 * dchain may be NULL (every chain on pool threads).  fokl_search_destroy sends back what is on order and waits for
 * everything in flight; the pool must outlive the search.
 */
typedef struct fokl_dchain fokl_dchain;         /* the device chain engine, declared further down */
typedef struct fokl_search fokl_search;
typedef struct fokl_spectrum fokl_spectrum;     /* one G2 job and its result */
typedef struct fokl_tape fokl_tape;             /* one model evaluation's noise */
typedef struct fokl_outcome fokl_outcome;       /* one model evaluation */
typedef struct fokl_search_params {
    int64_t n;                                  /* rows of the whole dataset (FR:1508: astar) */
    double a, b, atau, btau;                    /* FR:1322-1348 */
    double threshav, threshstda, threshstdb;    /* FR:1670-1671 */
    double guess_margin;                        /* decisions from the least-squares intercept: relative distance kept */
    int32_t draws;                              /* burnin + draws: iterations per chain */
    int32_t half0;                              /* ceil(draws / 2): first row of the intercept statistic (FR:1671) */
    int32_t aic;                                /* FR:1653-1654 */
    int32_t lookahead, foresight;               /* G2 jobs ahead of the tests; tests left when the next model is foreseen */
    int32_t speculation_max;                    /* tapes on order at most */
    int32_t tentative_tapes, test_rewinds;      /* 0: no tape is ordered ahead; tests: a discarded tape before every order */
    int32_t device_chain_columns;               /* device chains for models of up to this many columns */
    int32_t finish_threads;                     /* of the pool (0: chains complete their normals themselves) */
    int32_t flip_guess;                         /* tests: the n-th guessed decision is taken wrong */
    int32_t device_rows;                        /* the pool's stream leaves its pre-states with dchain: kill tests' tapes
                                                   stay rows, the device expands them (fokl_dchain_submit_rows) */
} fokl_search_params;
int fokl_search_create(fokl_host_pool *pool, fokl_dchain *dchain, const fokl_search_params *params, fokl_search **out);
/* G2 on the device engine (fokl_dspectral_*, declared further down; NULL: everything back to the pool's LAPACK threads)
 * for models of up to max_columns columns whose result is not wanted before the device can have it: a job requested
 * `slack` times the kernels' duration (or more) ahead of its kill test goes to the device, the others to the pool;
 * slack = 0: all of them, < 0: the default (1.5).  lookahead: how many tests ahead device jobs are requested (< 0: the
 * default, 32; the pool's jobs stay at fokl_search_params.lookahead).  Between fokl_search_hold_spectral(s, 1) and (s, 0)
 * fokl_search_spectral only stages its jobs; the closing call launches them as one grid. */
typedef struct fokl_dspectral fokl_dspectral;
int fokl_search_bind_spectral(fokl_search *search, fokl_dspectral *engine, int max_columns, double slack, int lookahead);
int fokl_search_hold_spectral(fokl_search *search, int hold);
/* G2 of the kill tests' models from the eigenpairs of the model each is tested against (fokl_pool_submit_spectral_update:
 * secular equation + one product, 4-5 x cheaper than a decomposition): for parents of from_columns columns or more (0, the
 * default: never) and at most `depth` such steps away from a fresh decomposition -- a step runs when the one before it has,
 * so `depth` cuts the chain of accepted tests along the predicted path into pieces the spectral threads work on side by side.
 * A chain of derivations advances slower than the loop tests and only the pieces inside the look-ahead window run side by
 * side: `lookahead` (0: fokl_search_params.lookahead) is the window's depth while derivation is on, in sub-stages whose model
 * has fewer than 192 columns.  Statistic 'spectral_updated' counts the models that were derived this way. */
int fokl_search_set_update(fokl_search *search, int from_columns, int depth, int lookahead);
/* How a kill test's BIC (FR:1686: `evtest < evmin`) is decided.  mode 0: from G2 of the trial model -- the loop waits for the
 * eigenpairs of every model it tests (round 4).  mode 1: from the sub-stage model's least-squares fit with the tested columns
 * removed one by one (sum of squared residuals without column c = SSR + b_c^2 / [(X'X)^-1]_cc, (X'X)^-1 and b one rank-one
 * downdate per accepted test): microseconds per test on the search thread; G2 is then requested for ACCEPTED models only,
 * feeds nothing but their chains (started when it arrives) and brings a second BIC (Gram identity on the eigenpairs'
 * betahat) that must agree with the decision's to `tolerance` (relative; <= 0: 1e-9) -- else the search ends as after a
 * mispredicted guess (fokl_search_mispredicted) and the driver repeats it in mode 0.  Sub-stages whose model is
 * numerically singular (smallest eigenvalue <= 1e-9 of the largest) run in mode 0 whatever is set here.  Statistics
 * 'direct_tests', 'direct_max_rel' (largest relative difference seen), 'chains_cancelled' (accepted models replaced
 * before anything looked at their draws: their chains never run). */
int fokl_search_set_decide(fokl_search *search, int mode, double tolerance);
/* Several ranks repeat this search side by side (rows or candidates sharded over GPUs) and must take every decision alike:
 * on != 0 keeps the arrival time of a chain's statistics out of every decision -- a second clause of FR:1670 that can be
 * guessed from the least-squares intercept IS guessed (a function of the all-reduced / all-gathered Gram alone), and a guess
 * that its chain does not confirm ends the search in the blocking fokl_search_verify at its end, where every rank finds it. */
int fokl_search_set_deterministic(fokl_search *search, int on);
void fokl_search_destroy(fokl_search *search);
const char *fokl_search_error(const fokl_search *search);
/* 1 after a guessed decision was not confirmed by its chain (the driver repeats the search without device chains) */
int fokl_search_mispredicted(const fokl_search *search);
int fokl_search_set_substage(fokl_search *search, const int64_t *term_ids, int columns);
/* the models the stream will probably serve next, in order (sizes in columns; is_model: a sub-stage's model) */
int fokl_search_speculate(fokl_search *search, const int32_t *sizes, const int32_t *is_model, int count);
int fokl_search_drop_speculation(fokl_search *search);
/* G2 (fokl_pool_submit_spectral) with the result buffer owned by the search: lamb [p1] | qty [p1] | betahat [p1] |
 * Qt [p1, p1] | moments [2]; gram must stay alive until the job has run */
int fokl_search_spectral(fokl_search *search, const double *gram, int ld, const int32_t *idx, int p1, fokl_spectrum **out);
/* The same for the model that is `parent`'s (a spectrum of this search, waited for or not) without its column number
 * parent_pos (NULL / -1: as fokl_search_spectral): derived from the parent's eigenpairs where fokl_search_set_update allows. */
int fokl_search_spectral_from(fokl_search *search, const double *gram, int ld, const int32_t *idx, int p1,
                              fokl_spectrum *parent, int parent_pos, fokl_spectrum **out);
int fokl_spectrum_done(fokl_spectrum *spectrum);
int fokl_spectrum_wait(fokl_search *search, fokl_spectrum *spectrum, const double **buffer, int *p1);
int fokl_spectrum_retain(fokl_search *search, fokl_spectrum *spectrum);     /* one more reference (released as below) */
void fokl_spectrum_release(fokl_search *search, fokl_spectrum *spectrum);
int fokl_search_model_begin(fokl_search *search, const double *gram, int ld, const int32_t *idx, int p1,
                            fokl_spectrum *given, const int32_t *then_sizes, const int32_t *then_model, int then_count,
                            fokl_spectrum **spectrum_out, fokl_tape **tape_out);
int fokl_search_model_commit(fokl_search *search, fokl_spectrum *spectrum, fokl_tape *tape, double dtd, int new_terms,
                             fokl_outcome **out);
int fokl_search_score(fokl_search *search, fokl_outcome *outcome, double sum_r, double sum_r2, int n_prev, int kill,
                      double *ev);
typedef struct fokl_outcome_view {
    const double *spectrum;                     /* lamb | qty | betahat | Qt | moments of the model */
    const int32_t *idx;                         /* its active-column indices [p1] */
    double ev, siglik, intercept_scale;         /* intercept_scale: NaN while unknown */
    int32_t p1, on_device;
} fokl_outcome_view;
int fokl_outcome_info(fokl_search *search, fokl_outcome *outcome, fokl_outcome_view *view);
int fokl_outcome_spectrum(fokl_search *search, fokl_outcome *outcome, fokl_spectrum **out);
int fokl_outcome_chain_ready(fokl_outcome *outcome);
int fokl_outcome_draws(fokl_search *search, fokl_outcome *outcome, const double **w);
int fokl_outcome_intercept_scale(fokl_search *search, fokl_outcome *outcome, double *scale);
/* FR:1656-1658 for the active columns `cols` of the outcome's model, from its draws in the eigenbasis (waits for the chain):
 * mean_abs[c] = |mean over rows half1 .. of beta_c|, rel_std[c] = std over rows half1 .. / |mean over rows half0 ..|. */
int fokl_outcome_new_term_stats(fokl_search *search, fokl_outcome *outcome, const int32_t *cols, int count, int half0,
                                int half1, double *mean_abs, double *rel_std);
void fokl_outcome_release(fokl_search *search, fokl_outcome *outcome);
void fokl_outcome_drop(fokl_search *search, fokl_outcome *outcome);
int fokl_search_verify(fokl_search *search, int block);
int fokl_search_register_forecast(fokl_search *search, const int32_t *key, int key_count, fokl_spectrum *spectrum,
                                  double dtd);
void fokl_search_clear_forecasts(fokl_search *search);
int fokl_search_likely_first_tests(fokl_search *search, fokl_spectrum *spectrum, int n_new, double siglik,
                                   int32_t *columns_out, int32_t *accepted_out, int *count);
/* counters / seconds in the order of csrc/fokl_search.cpp's Stat enumeration (-> their number); the trace: 5 doubles per
 * evaluation (columns, built, ev, kill, the mean intercept draw over rows half0 .. of that evaluation's chain -- FR:1671's
 * scale before the abs() -- or NaN where the search never looked at that chain's statistics) */
int fokl_search_stats(const fokl_search *search, double *values, int count);
int64_t fokl_search_trace(const fokl_search *search, double *records, int64_t capacity);
/*
 * One sub-stage's kill tests.  gram [(active + 1)^2]: Gram of the active columns with y last; columns / mean_abs /
 * rel_std [proposals]: the new terms in testing order (ascending |mean beta|, FR:1663-1664): active-column index,
 * |mean beta| (FR:1656), std / |mean| (FR:1657-1658); slots [active]: device slot of every active column (the key of
 * forecasts); best: the sub-stage's model.  ahead_*: G2 jobs the caller submitted for first trial sets (keys: active
 * column indices, CSR offsets).  vm_next: columns the coming sub-stage adds (-1: there is none).  Callbacks (may be
 * NULL) run on the calling thread: foresee(predicted kill set) towards the end of the loop; idle_work once, when the
 * first test's tape and G2 are under way (or at the end); residual: sum r, sum r^2 of y - X betahat for a candidate
 * that (nearly) interpolates the data.  killed [>= proposals] receives the kill set (ascending); best: the model the
 * sub-stage ends on (best_is_new: a new handle the caller owns, else the one passed in).
 */
typedef struct fokl_kill_tests_args {
    const double *gram;
    const int32_t *columns;
    const double *mean_abs, *rel_std;
    const int32_t *slots;
    fokl_outcome *best;
    const int32_t *ahead_keys, *ahead_offsets;
    fokl_spectrum *const *ahead_spectra;
    void *user;
    void (*foresee)(void *user, const int32_t *killed, int count);
    int (*idle_work)(void *user);
    int (*residual)(void *user, const int32_t *idx, int p1, const double *betahat, double *sum_r, double *sum_r2);
    int32_t active, proposals, n_prev, vm_next, ahead_count;
} fokl_kill_tests_args;
typedef struct fokl_kill_tests_result {
    int32_t *killed;
    fokl_outcome *best;
    double evmin;
    int32_t killed_count, best_is_new;
} fokl_kill_tests_result;
int fokl_search_kill_tests(fokl_search *search, const fokl_kill_tests_args *args, fokl_kill_tests_result *result);

/* ------------------------------------------------------------------------------------------------------ */
/* The sub-stage loop next to the kill-test loop (round 6; csrc/fokl_run.cpp).  Replaces the bookkeeping of    */
/* FR:1602-1748 around the calls above: enumeration, build-ahead, model evaluation, statistics, stop rule.     */
/* ------------------------------------------------------------------------------------------------------ */
/* The device as a table of entry points with the signatures of fokl_hip.h (ctx is handed back as their first
 * argument): the library's own functions on a fokl_ctx, or a checker backend's callbacks (CPU tests). */
typedef struct fokl_backend_ops {
    void *ctx;
    int (*reserve_slots)(void *ctx, int n_slots);
    int (*build_terms)(void *ctx, const int32_t *terms, int T, const int32_t *slots);
    int (*gram)(void *ctx, const int32_t *row_slots, int nr, const int32_t *col_slots, int nc, double *out, int path,
                int allreduce);
    int (*gram_launch)(void *ctx, const int32_t *row_slots, int nr, const int32_t *col_slots, int nc, int allreduce);
    int (*gram_fetch)(void *ctx, double *out, int64_t count);
    int (*gram_ready)(void *ctx);                                       /* may be NULL */
    int (*bic_resid)(void *ctx, const int32_t *slots, int nc, const double *betahat, double *out, int allreduce);
    int (*bic_resid_launch)(void *ctx, const int32_t *slots, int nc, const double *betahat);
    int (*bic_resid_fetch)(void *ctx, double *out, int allreduce);
    int (*bic_resid_terms_launch)(void *ctx, const int32_t *terms, int n_terms, const double *betahat);   /* may be NULL */
    int32_t kernel_id;                                                  /* FOKL_KERNEL_* of the uploaded dataset */
} fokl_backend_ops;
typedef struct fokl_run_params {
    int32_t m, n_phis;                          /* inputs; orders the kernel offers (FR:1747) */
    int32_t way3, tolerance, gimmie;            /* FR:208-212 */
    int32_t draws, half0;                       /* iterations per chain; first row of the intercept statistic */
    int32_t lookahead, lookahead_native, foresight, speculate_across;   /* engine.py's knobs of the same names */
    int32_t forecast_early, forecast_polls;
    int32_t matrix_free;                        /* K3 without the stored columns where the terms allow it */
    int32_t update_from, update_depth, update_lookahead;   /* fokl_search_set_update's arguments (update_depth 0: never set) */
    int32_t head_start;                         /* fokl_run_create launches the first sub-stage's K1 + K2 */
    int32_t slot_capacity;                      /* device column slots to start with */
} fokl_run_params;
#define FOKL_RUN_STATS 16
typedef struct fokl_run fokl_run;
/* Seed Gram + (head_start) the first sub-stage's columns and Gram block under way: before the caller creates its pool. */
int fokl_run_create(const fokl_backend_ops *ops, const fokl_run_params *params, fokl_run **out);
/* fokl_search_set_update's arguments as the caller configured its search (the loop shortens the derivation depth in the
 * sub-stage after which the stop rule may end the search; depth 0: never touched). */
int fokl_run_set_update(fokl_run *run, int from_columns, int depth, int lookahead);
/* The loop, on `search` (fokl_search_create on the caller's pool; decisions, update depth etc. configured by the caller). */
int fokl_run_search(fokl_run *run, fokl_search *search);
/* What it found: rows of the interaction matrix, length of the BIC trace, sub-stages; the outcome handles of the returned
 * model and of the last sub-stage's survivor (the caller owns them: fokl_outcome_drop; they may be one and the same). */
int fokl_run_result(const fokl_run *run, int32_t *mtx_rows, int32_t *evs_count, int32_t *substages,
                    fokl_outcome **best_model, fokl_outcome **last_model);
/* mtx [mtx_rows][m], evs, per sub-stage the number of new terms and -- concatenated -- their |mean beta| and std / |mean|
 * (FR:1656-1658), stats [FOKL_RUN_STATS]: columns built, sub-stages, forecasts used / early, matrix-free residual passes,
 * seconds waiting for K3, seconds by phase (prepare, model, statistics, tests, wrap-up).  NULL: skipped.  -> length of the
 * statistics arrays */
int fokl_run_arrays(const fokl_run *run, int32_t *mtx, double *evs, int32_t *stat_sizes, double *stat_mean_abs,
                    double *stat_rel_std, double *stats);
const char *fokl_run_error(const fokl_run *run);
void fokl_run_destroy(fokl_run *run);

/* ------------------------------------------------------------------------------------------------------ */
/* G3 on the device: finishing of the polar normals + the D-iteration recursion (FoKLRoutines.py:1519-1548)  */
/* ------------------------------------------------------------------------------------------------------ */

/* A device-chain engine on HIP device `device`: a dispatcher thread, a few streams, `slots` chains that may be
 * alive (in flight, or finished with their draws still in device memory) at a time.  The random stream stays on
 * the host (fokl_noise_tape / the pool's noise thread); what is submitted here is the arithmetic on a tape:
 * the sqrt(-2 log r2 / r2) half of the polar method and the recursion in the eigenbasis that fokl_gibbs_chain
 * runs on the host -- same operations in the same order, so the draws differ only through log() (1e-16).
 *
 * fokl_dchain_submit queues a chain and returns at once.  The tape may still be on record: `progress` (may be
 * NULL: the tape is complete) is polled by the dispatcher until it reaches `draws`; a negative value fails the
 * job; `block_done` / `block` (may be NULL: the arrays are filled as soon as `progress` says so) are the flags of the
 * threads that materialise the tape (fokl_pool_submit_noise), polled the same way.  `finished` != 0: the normals are
 * final already (host finish threads), else they are raw pairs + `lead` as fokl_noise_tape leaves them.  lamb / qty are copied at submit; the tape's arrays must stay valid until fokl_dchain_poll
 * reports 1 or fokl_dchain_wait / fokl_dchain_release has returned.
 * fokl_dchain_wait sleeps until the chain has run and returns stats_out[4 + p1] = {bstar < 0 seen, last sigma^2,
 * last tau^2, rows averaged, mean over rows stat_first .. draws - 1 of w} -- what the kill tests look at
 * (FR:1671: mean intercept draw = mean w . Q[0, :]).  fokl_dchain_fetch_w copies the draws in the eigenbasis
 * w [draws, p1] (betas = w Q') to the host; fokl_dchain_release frees the slot (idempotent).
 * `stats_area` (may be NULL) receives the address of the job's statistics in page-locked host memory: the five + p1
 * doubles the recursion kernel writes -- the four + p1 of fokl_dchain_wait, then the job's ticket (as a double),
 * stored last with system-wide release semantics: a caller may poll that word instead of calling fokl_dchain_poll;
 * one more double behind it holds the seconds the chain's wavefront ran (the kernel's own clock).
 * The area belongs to the job's slot: valid until the job is released.
 * Errors: FOKL_ERR_STATE when every slot is taken (the caller runs that chain on the host). */
int fokl_dchain_create(int device, int slots, fokl_dchain **out);
void fokl_dchain_destroy(fokl_dchain *engine);
int fokl_dchain_submit(fokl_dchain *engine, int p1, int draws, const double *lamb, const double *qty, double b,
                       double btau, double dtd, double sigsqd0, double tausqd0, const double *normals,
                       const int32_t *lead, const double *gam_sig, const double *gam_tau, const int32_t *progress,
                       const int32_t *block_done, int block, int finished, int stat_first, int64_t *ticket,
                       const double **stats_area);
/*
 * The tape as ROWS (round 4): the engine keeps its own copy of the random stream -- fokl_dchain_prestate_ring hands out the
 * page-locked ring a stream created with it (fokl_stream_create / fokl_pool_create) leaves its pre-states in, one
 * workgroup per segment regenerates MT19937 -> tempering -> numpy's doubles -> x = 2 d - 1 from a pre-state into a ring in
 * device memory (when a chain first needs the segment), fokl_dchain_bind_stream says whose pre-states the ring holds -- and
 * fokl_dchain_submit_rows expands a tape's 32-byte rows there: accepted attempts re-decided from x1^2 + x2^2 (the host's
 * roundings: same flags), normals finished, gamma variates formed.  What crosses the bus per chain is its rows; the
 * arithmetic differs from the host's expansion only through log().
 */
int fokl_dchain_prestate_ring(fokl_dchain *engine, uint32_t **ring, int *entries);
int fokl_dchain_bind_stream(fokl_dchain *engine, const fokl_stream *stream);
int fokl_dchain_submit_rows(fokl_dchain *engine, int p1, int draws, const double *lamb, const double *qty, double b,
                            double btau, double dtd, double sigsqd0, double tausqd0, double astar, double atau_star,
                            const fokl_tape_row *rows, const double *gam_sig, const double *gam_tau,
                            const int32_t *progress, const uint64_t *span, int stat_first, int64_t *ticket,
                            const double **stats_area);
/* segments regenerated on the device so far, chains submitted as rows */
int fokl_dchain_stream_stats(fokl_dchain *engine, int64_t *segments_made, int64_t *rows_jobs);
int fokl_dchain_poll(fokl_dchain *engine, int64_t ticket);
int fokl_dchain_flush(fokl_dchain *engine);   /* issue what is queued now: nothing more is coming for a while */
int fokl_dchain_wait(fokl_dchain *engine, int64_t ticket, double *stats_out);
int fokl_dchain_fetch_w(fokl_dchain *engine, int64_t ticket, double *w_out);
int fokl_dchain_release(fokl_dchain *engine, int64_t ticket);
/* The same without waiting: 1 = the slot is free (now or before), 0 = the chain has not run yet. */
int fokl_dchain_try_release(fokl_dchain *engine, int64_t ticket);
/* seconds the dispatcher spent issuing work, number of chains issued, number of recursion launches (chains whose
 * tapes are ready together go out as one launch: FOKL_DCHAIN_BATCH chains or FOKL_DCHAIN_DELAY_US after the oldest was
 * queued, at once when somebody waits for a result); `staged` = chains whose tape was not in page-locked memory and
 * went through copy calls + a device staging buffer instead of being read in place */
int fokl_dchain_stats(fokl_dchain *engine, double *busy_seconds, int64_t *issued, int64_t *launches, int64_t *staged);

/* ---- G2 on the device: eigen-decompositions of candidate models' XtX sub-blocks -------------------------------------
 * Replaces, for models of up to FOKL_DSPECTRAL_MAX_COLUMNS columns, the scipy.linalg.eigh call of FoKLRoutines.py:1499
 * and the products of FR:1502-1504 that hang on it (on the host: fokl_pool_submit_spectral, LAPACK dsyevr on a thread).
 * One workgroup per matrix runs a cyclic Jacobi iteration on the matrix in LDS, one wavefront per eigenvector column
 * replays its rotations (csrc/fokl_spectral_device.inc).
 *
 * fokl_dspectral_submit copies the sub-block XtX[idx][idx], Xty = gram[idx][ycol], the ones row gram[0][idx] and
 * gram[0][ycol], gram[ycol][ycol] (gram: [ld][ld] row-major, symmetric, column 0 the ones column) into page-locked memory
 * of the engine -- the caller's array is not referenced after the call -- and stages the job; launch != 0 launches what is
 * staged at once, 0 leaves it for fokl_dspectral_flush (several jobs become one grid) or for the first poll / wait of any
 * of them.  *result is the job's page-locked result area, owned by the engine until fokl_dspectral_release:
 *   lamb [p1] ascending | qty = Q'Xty [p1] | betahat [p1] | Qt [p1][p1] (row j = eigenvector j) | sum r, sum r^2 |
 *   sweeps, rotations, seconds on the device, 1 if the sweeps did not converge | the job's ticket (as a double)
 * the ticket stored last with system-wide release semantics: a caller may poll that word.  Eigenvector signs follow
 * engine.eigh_canonical (largest-magnitude component positive, first on ties) unless fokl_dspectral_set_signs(e, 0).
 * fokl_dspectral_poll: 1 = has run, 0 = not yet, < 0 = -error.  fokl_dspectral_wait: FOKL_OK, or FOKL_ERR_NUMERIC when
 * the sweeps did not converge.  fokl_dspectral_release waits for a job still in flight; idempotent. */
#define FOKL_DSPECTRAL_MAX_COLUMNS 192
int fokl_dspectral_create(int device, fokl_dspectral **out);
void fokl_dspectral_destroy(fokl_dspectral *engine);
int fokl_dspectral_max_columns(void);
int fokl_dspectral_set_signs(fokl_dspectral *engine, int canonical);
int fokl_dspectral_submit(fokl_dspectral *engine, const double *gram, int ld, const int32_t *idx, int p1, int ycol,
                          int launch, int64_t *ticket, double **result);
int fokl_dspectral_flush(fokl_dspectral *engine);
int fokl_dspectral_poll(fokl_dspectral *engine, int64_t ticket);
int fokl_dspectral_wait(fokl_dspectral *engine, int64_t ticket);
int fokl_dspectral_release(fokl_dspectral *engine, int64_t ticket);
int fokl_dspectral_stats(fokl_dspectral *engine, int64_t *submitted, int64_t *launches);
/* Page-locked host memory for tapes: the device reads such a tape in place (no copy calls on the dispatcher). */
int fokl_host_alloc(size_t bytes, void **out);
int fokl_host_free(void *ptr);

#ifdef __cplusplus
}
#endif
#endif /* FOKL_HIP_INTERNAL_H */
