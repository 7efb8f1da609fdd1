// Host threads of one fit (G2/G3 of SURVEY 8(a) are N-independent and latency bound, so they stay on the host; this
// file keeps them off the Python driver thread).  Four kinds of work, each on its own queue(s):
//
//   noise    ONE thread that walks the numpy legacy random stream for the lifetime of the pool and records noise tapes
//            strictly in submission order -- since round 4 as ROWS OF POSITIONS (fokl_stream_walk): the stream's words,
//            doubles and accept flags come from the stream's bulk threads (csrc/fokl_stream.cpp), the walk only counts
//            flags and decides the two gamma draws per iteration;
//   finish   threads that turn a tape's rows into its numbers (fokl_stream_expand: the accepted pairs, the leading and
//            trailing normals, the gamma variates) and, where a host chain will read it, complete the normals in place
//            (the log / sqrt half of the polar method) -- all of them on every tape, block-interleaved, while it is still
//            being walked;
//   chain    threads that turn a tape into the draws of one candidate (the sequential recursion FR:1521-1548),
//            following the finished blocks;
//   spectral threads that diagonalise a candidate's XtX sub-block: LAPACK dsyevr exactly as scipy.linalg.eigh calls
//            it (FoKLRoutines.py:1499; the function pointer is scipy's own, handed in by the Python side), the sign
//            convention of engine.eigh_canonical, Q'Xty and betahat (FR:1502-1504).  These carry no random numbers,
//            so the driver may submit them speculatively for models it might evaluate next.
//
// Buffers named in a job belong to the caller and must stay alive until fokl_pool_wait returned for that job.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/prctl.h>

#include "../../include/fokl_hip_internal.h"
#include "fokl_spin.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip

// fokl_sampler.cpp (library-internal)
extern "C" __attribute__((visibility("hidden"))) void fokl_finish_tape_rows(int p1, double *normals, const double *pair_r2,
                                                                            const int32_t *lead, int k0, int k1);
extern "C" __attribute__((visibility("hidden"))) int fokl_gibbs_chain_from_raw_blocks(
    const double *lamb, const double *qty, int p1, double b, double btau, double dtd, double sigsqd0, double tausqd0,
    int draws, const double *normals, const double *pair_r2, const int32_t *lead, const double *gam_sig,
    const double *gam_tau, const int32_t *block_done, int block, double *w_out, int32_t *bstar_negative);

// CPU time of the library's own threads by kind, process-wide, added when a thread ends (a fit's pool lives as long as the
// fit): what a fit costs in CPU-seconds and where -- the figure that bounds fits side by side on a host with a CPU quota.
// kinds: 0 walker, 1 chain, 2 finish, 3 spectral, 4 bulk (fokl_stream.cpp), 5 device-chain dispatcher (read live)
static std::atomic<int64_t> g_thread_cpu_ns[8];

extern "C" __attribute__((visibility("hidden"))) void fokl_note_thread_cpu(int kind)
{
    timespec ts;
    if (kind >= 0 && kind < 8 && clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0)
        g_thread_cpu_ns[kind].fetch_add((int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec, std::memory_order_relaxed);
}

extern "C" __attribute__((visibility("hidden"))) int64_t fokl_dchain_dispatcher_cpu_ns();   // fokl_chain_device.inc / host-only stub

extern "C" int fokl_thread_cpu_seconds(double *seconds, int count)
{
    if (!seconds || count < 6) {
        fokl_set_global_error("fokl_thread_cpu_seconds: room for six values is needed");
        return FOKL_ERR_ARG;
    }
    for (int k = 0; k < 5; ++k) seconds[k] = 1e-9 * (double)g_thread_cpu_ns[k].load(std::memory_order_relaxed);
    seconds[5] = 1e-9 * (double)fokl_dchain_dispatcher_cpu_ns();
    return 6;
}

namespace {

struct ThreadCpuNote {                                      // at the end of a thread's function
    int kind;
    ~ThreadCpuNote() { fokl_note_thread_cpu(kind); }
};

using dsyevr_fn = void (*)(char *jobz, char *range, char *uplo, int *n, double *a, int *lda, double *vl, double *vu,
                           int *il, int *iu, double *abstol, int *m, double *w, double *z, int *ldz, int *isuppz,
                           double *work, int *lwork, int *iwork, int *liwork, int *info);

using dsyevd_fn = void (*)(char *jobz, char *uplo, int *n, double *a, int *lda, double *w, double *work, int *lwork,
                          int *iwork, int *liwork, int *info);

using dgemm_fn = void (*)(char *transa, char *transb, int *m, int *n, int *k, double *alpha, double *a, int *lda, double *b,
                          int *ldb, double *beta, double *c, int *ldc);

enum class Kind { noise, chain, finish, spectral };

struct Queue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<fokl_host_job *> q;
    bool stop = false;
};

}  // namespace

struct fokl_host_job {
    void (*then)(void *) = nullptr;         // chain jobs: run by the chain thread behind a chain that succeeded
    void *then_arg = nullptr;
    Kind kind;
    fokl_host_pool *pool = nullptr;
    std::atomic<int> done{0};
    int status = FOKL_OK;
    std::string error;
    // noise: a tentative tape is recorded ahead of the decision that it is needed; the driver then commits it (+1) or
    // aborts it (-1: the stream is put back where the tape started)
    bool tentative = false;
    std::atomic<int> verdict{0};
    int64_t t_submit = 0, t_start = 0, t_recorded = 0;      // FOKL_POOL_TRACE
    // a noise job is done when the stream is through with it AND the finish jobs submitted with it have left the tape
    std::atomic<int> pending{1};
    // noise: the walker has begun the tape (or sent it back unwalked); the finish jobs submitted with it sleep until then
    // (fokl_host_pool::start_cv) -- a model's tape is queued behind a sub-stage's worth of kill-test tapes, milliseconds
    // during which two threads per tape used to poll its progress word every 10 us
    std::atomic<int> started{0};
    fokl_host_job *parent = nullptr;        // finish job submitted with a noise job: that job
    // noise / chain
    int p1 = 0, draws = 0;
    double astar = 0, atau_star = 0;
    fokl_tape_row *rows = nullptr;          // noise / finish: the tape as the walk leaves it
    uint64_t hold = 0;                      // noise: the stream is kept readable from here on until the job settles
    bool held = false;
    uint64_t *span_out = nullptr;           // noise: [hold position, walker position behind the tape]; the hold then is
                                            // the caller's to release (fokl_pool_release_hold)
    bool rows_only = false;                 // noise: nobody materialises the tape here (the device expands the rows)
    bool finish_normals = false;            // finish: complete the normals in place after expanding them
    double *normals = nullptr, *pair_r2 = nullptr, *gam_sig = nullptr, *gam_tau = nullptr;
    int32_t *lead = nullptr, *progress = nullptr;
    // chain
    const double *lamb = nullptr, *qty = nullptr;
    double b = 0, btau = 0, dtd = 0, sigsqd0 = 0, tausqd0 = 0;
    double *w_out = nullptr;
    int32_t *bstar_negative = nullptr;
    // finish (internal, freed by the thread that ran it) and chain on a tape finished by those
    const int32_t *raw_block_done = nullptr; // chain on a tape whose blocks are expanded, not finished, by the finish threads
    int32_t *block_done = nullptr;
    int block = 0, part = 0, parts = 0;
    bool self_owned = false;
    // spectral
    const double *gram = nullptr;
    int ld = 0, ycol = 0;
    std::vector<int32_t> idx;
    double *lamb_out = nullptr, *qt_out = nullptr, *qty_out = nullptr, *betahat_out = nullptr, *moments_out = nullptr;
    // spectral, from the eigenpairs of the model with one more column (fokl_pool_submit_spectral_update)
    const double *parent_lamb = nullptr, *parent_qt = nullptr;
    int parent_pos = -1;                    // which of the parent's columns this model lacks
    bool parent_failed = false;
    int32_t *updated_out = nullptr;         // 1 = derived from the parent, 0 = decomposed afresh after all
    // jobs that read this job's eigenpairs: queued (at the front) when it has run.  Under fokl_host_pool::dep_m
    std::vector<fokl_host_job *> dependents;
    bool has_run = false;
    int run_status = FOKL_OK;
};

struct fokl_host_pool {
    Queue noise_q, chain_q, spectral_q;
    std::deque<Queue> finish_q;             // one per finish thread: every tape is split over all of them
    std::vector<std::thread> threads;
    std::vector<size_t> spectral_thread_ids;    // indices into `threads` (fokl_pool_spectral_affinity)
    dsyevr_fn dsyevr = nullptr;
    // LAPACK's divide-and-conquer driver for the wider models (fokl_pool_use_dsyevd): same tridiagonal reduction as dsyevr,
    // eigenpairs within ~3e-12 of dsyevr's in the chain's noise map (profiles/eigh_drivers_r04.txt), 1.3-1.5 x faster from
    // 80 columns on
    dsyevd_fn dsyevd = nullptr;
    int dsyevd_from = 0;
    dgemm_fn dgemm = nullptr;                   // fokl_pool_use_dgemm: the product of the eigen-update
    std::mutex dep_m;                           // fokl_host_job::dependents / has_run of every spectral job
    // the random stream: walked by the noise thread, produced by the stream's own bulk threads; the caller's state
    // (mt_key ...) is read at creation and written back when the pool is destroyed
    fokl_stream *stream = nullptr;
    uint32_t *mt_key = nullptr;
    int32_t *mt_pos = nullptr, *has_gauss = nullptr;
    double *gauss_cache = nullptr;
    std::atomic<int64_t> noise_busy_ns{0}, chain_busy_ns{0}, finish_busy_ns{0}, spectral_busy_ns{0};
    // where the serial resource waits: for the next request (empty queue) and for the verdict on a tentative tape
    std::atomic<int64_t> noise_queue_wait_ns{0}, noise_verdict_wait_ns{0};
    // completion of any job (fokl_pool_wait spins briefly, then sleeps here: a fit must not burn a core per waiter --
    // eight ranks may share a CPU quota far below eight times the thread count)
    std::mutex done_m;
    std::condition_variable done_cv;
    std::mutex start_m;                         // fokl_host_job::started of every noise job
    std::condition_variable start_cv;
    // FOKL_POOL_TRACE=<file>: one line per noise job (steady-clock ns: submitted, started, recorded, verdict seen; p1;
    // tentative; verdict), appended when the pool is destroyed -- tools/pool_trace.py lines it up with the driver's log
    std::string trace_path;
    std::vector<std::array<int64_t, 7>> trace;
    // FOKL_POOL_TEST_DELAY_US (tests only): the noise thread sleeps this long between looking at the verdicts of its open
    // tapes and taking the next request -- the window in which a driver can send tapes back and queue a new one
    int test_delay_us = 0;
    // FOKL_EIGH_SIGNS=lapack: eigenvectors keep the signs dsyevr returns (the untouched reference's draws on a host
    // whose BLAS forms the same Gram); default: largest-magnitude component positive (what the goldens pin)
    bool lapack_signs = false;
    // A (nearly) singular XtX (smallest eigenvalue <= singular * largest; FOKL_EIGH_SINGULAR, default 1e-9) has no
    // eigenvectors to speak of in its (near) null space: what a fit then selects is whatever the reference's own driver
    // returns there (random problems with a third as many terms as rows: tests/stress/random_parity.py 353, 465, 494, 584
    // select other models with dsyevd or derived eigenpairs, the reference's with dsyevr).  Such models are decomposed by
    // dsyevr, as scipy.linalg.eigh does (FR:1499) -- neither dsyevd nor the update; the BASELINE configurations' models
    // (condition numbers 1e5 .. 1e8) are not among them.
    double singular = 1e-9;
};

namespace {

inline int64_t now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

void finish(fokl_host_job *job, int status, const char *what)
{
    fokl_host_pool *pool = job->pool;                       // the job may be freed by its waiter right after `done`
    job->status = status;
    if (status != FOKL_OK) job->error = what;
    {
        std::lock_guard<std::mutex> lock(pool->done_m);
        job->done.store(1, std::memory_order_release);
    }
    pool->done_cv.notify_all();
}

void spectral_tail(fokl_host_pool *pool, fokl_host_job *job, const double *xty);

// XtX sub-block -> (lamb, Q', Q'Xty, betahat).  Column-major copy + uplo 'L', abstol 0, range 'A', workspace from a
// query: the call scipy.linalg.eigh(XtX) makes (driver 'evr'), so the eigenpairs are the reference's bit for bit.
int spectral(fokl_host_pool *pool, fokl_host_job *job, std::string &err)
{
    const int n = (int)job->idx.size();
    const int32_t *idx = job->idx.data();
    const double *g = job->gram;
    const size_t ld = (size_t)job->ld;
    // scratch of the calling spectral thread, kept from job to job (a 140-column block is 157 KB: fresh from malloc that is
    // an mmap, 40 page faults and a munmap per job)
    static thread_local std::vector<double> a, xty, work;
    static thread_local std::vector<int> iwork, isuppz;
    if (a.size() < (size_t)n * n) a.resize((size_t)n * n);
    if (xty.size() < (size_t)n) xty.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        const double *row = g + (size_t)idx[i] * ld;
        for (int j = 0; j < n; ++j) a[(size_t)j * n + i] = row[idx[j]];       // a(i, j) column-major = XtX[i][j]
        xty[i] = row[job->ycol];
    }
    char jobz = 'V', range = 'A', uplo = 'L';
    int nn = n, lda = n, ldz = n, il = 1, iu = n, m = 0, info = 0, lwork = -1, liwork = -1, iwork_query = 0;
    double vl = 0.0, vu = 1.0, abstol = 0.0, work_query = 0.0;
    double *z = job->qt_out;                                // z(i, j) at z[j * n + i]: row j of Q' = eigenvector j
    bool by_dsyevd = false;
    if (pool->dsyevd && pool->dsyevd_from > 0 && n >= pool->dsyevd_from && !pool->lapack_signs) {
        // divide and conquer: the eigenvectors overwrite the matrix, column j = eigenvector j -- the layout of Q' row-major
        std::memcpy(z, a.data(), sizeof(double) * (size_t)n * n);
        pool->dsyevd(&jobz, &uplo, &nn, z, &lda, job->lamb_out, &work_query, &lwork, &iwork_query, &liwork, &info);
        if (info != 0) {
            err = "dsyevd workspace query failed";
            return FOKL_ERR_NUMERIC;
        }
        lwork = (int)work_query;
        liwork = iwork_query;
        if (work.size() < (size_t)std::max(1, lwork)) work.resize((size_t)std::max(1, lwork));
        if (iwork.size() < (size_t)std::max(1, liwork)) iwork.resize((size_t)std::max(1, liwork));
        pool->dsyevd(&jobz, &uplo, &nn, z, &lda, job->lamb_out, work.data(), &lwork, iwork.data(), &liwork, &info);
        if (info != 0) {
            err = "dsyevd did not converge (info = " + std::to_string(info) + ")";
            return FOKL_ERR_NUMERIC;
        }
        by_dsyevd = true;
    }
    if (by_dsyevd && !(job->lamb_out[0] > pool->singular * job->lamb_out[n - 1])) {
        by_dsyevd = false;                                  // numerically singular: the reference's driver decides
        lwork = liwork = -1;
    }
    if (!by_dsyevd) {
        if (isuppz.size() < (size_t)2 * std::max(1, n)) isuppz.resize((size_t)2 * std::max(1, n));
        pool->dsyevr(&jobz, &range, &uplo, &nn, a.data(), &lda, &vl, &vu, &il, &iu, &abstol, &m, job->lamb_out, z, &ldz,
                     isuppz.data(), &work_query, &lwork, &iwork_query, &liwork, &info);
        if (info != 0) {
            err = "dsyevr workspace query failed";
            return FOKL_ERR_NUMERIC;
        }
        lwork = (int)work_query;
        liwork = iwork_query;
        if (work.size() < (size_t)std::max(1, lwork)) work.resize((size_t)std::max(1, lwork));
        if (iwork.size() < (size_t)std::max(1, liwork)) iwork.resize((size_t)std::max(1, liwork));
        pool->dsyevr(&jobz, &range, &uplo, &nn, a.data(), &lda, &vl, &vu, &il, &iu, &abstol, &m, job->lamb_out, z, &ldz,
                     isuppz.data(), work.data(), &lwork, iwork.data(), &liwork, &info);
        if (info != 0 || m != n) {
            err = "dsyevr did not converge (info = " + std::to_string(info) + ")";
            return FOKL_ERR_NUMERIC;
        }
    }
    spectral_tail(pool, job, xty.data());
    return FOKL_OK;
}

// Signs, Q'Xty, betahat and the residual moments from eigenpairs in job->lamb_out / job->qt_out.
void spectral_tail(fokl_host_pool *pool, fokl_host_job *job, const double *xty)
{
    const int n = (int)job->idx.size();
    const int32_t *idx = job->idx.data();
    const double *g = job->gram;
    const size_t ld = (size_t)job->ld;
    double *z = job->qt_out;
    // sign convention of engine.eigh_canonical: the largest-magnitude component (first one on ties) is positive
    for (int j = 0; j < n && !pool->lapack_signs; ++j) {
        double *v = z + (size_t)j * n;
        int piv = 0;
        double best = std::fabs(v[0]);
        for (int i = 1; i < n; ++i)
            if (std::fabs(v[i]) > best) {
                best = std::fabs(v[i]);
                piv = i;
            }
        if (v[piv] < 0.0)
            for (int i = 0; i < n; ++i) v[i] = -v[i];
    }
    // qty = Q'Xty, betahat = Q (qty / lamb)   (FR:1502-1504)
    double *qty = job->qty_out, *bh = job->betahat_out;
    for (int j = 0; j < n; ++j) {
        const double *v = z + (size_t)j * n;
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += v[i] * xty[i];
        qty[j] = s;
    }
    for (int i = 0; i < n; ++i) bh[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        const double *v = z + (size_t)j * n;
        const double c = qty[j] / job->lamb_out[j];
        for (int i = 0; i < n; ++i) bh[i] += v[i] * c;
    }
    if (job->moments_out) {
        // Residual moments of y - X betahat from the Gram alone (column 0 of gram is the ones column):
        //   sum r   = sum y - sum_j (1'x_j) b_j
        //   sum r^2 = y'y - 2 b'Xty + b'XtX b
        // The second cancels 3-4 digits at the signal-to-noise ratios of interest; accumulating in 80-bit extended
        // precision leaves the rounding of the Gram entries themselves as the only error (about 1e-12 relative).
        const double *ones = g;                                          // row 0 of gram
        long double s1 = ones[job->ycol], cross = 0.0L, quad = 0.0L;
        for (int i = 0; i < n; ++i) {
            const double *row = g + (size_t)idx[i] * ld;
            long double acc = 0.0L;
            for (int j = 0; j < n; ++j) acc += (long double)row[idx[j]] * bh[j];
            quad += acc * bh[i];
            cross += (long double)xty[i] * bh[i];
            s1 -= (long double)ones[idx[i]] * bh[i];
        }
        const long double yty = g[(size_t)job->ycol * ld + job->ycol];
        job->moments_out[0] = (double)s1;
        job->moments_out[1] = (double)(yty - 2.0L * cross + quad);
    }
}

// ---- G2 of a model from the eigenpairs of the model with one more column ------------------------------------------
// XtX' = XtX without row / column c.  With XtX = Q diag(lam) Q' and z = row c of Q, the eigenvalues of XtX' are the
// n - 1 roots mu_k of g(mu) = sum_j z_j^2 / (lam_j - mu), one strictly inside each (lam_k, lam_k+1), and its eigenvectors
// the rows != c of Q x_k with x_k[j] = z_j / (lam_j - mu_k), normalised.  Each root is found in the coordinate
// tau = mu - (the nearer pole), so that lam_j - mu = (lam_j - pole) - tau never cancels; the x_k are formed from the z^ for
// which the COMPUTED roots are exact (Gu & Eisenstat 1994: z^_j^2 = prod_k (mu_k - lam_j) / prod_{i != j} (lam_i - lam_j)),
// which keeps them orthogonal to working precision whatever the conditioning.  O(n^2) + one (n-1) x n x (n-1) product
// (BLAS dgemm) instead of a tridiagonal reduction: 0.2 ms instead of dsyevd's 0.8 at 140 columns.  Accuracy, chained over 40
// deletions on Gram matrices of Bernoulli terms: tests/stress/eigen_deletion_study.py (no growth; within 5x of LAPACK's
// own distance from the exact eigenpairs in the chain's noise map).  Anything doubtful -- a (nearly) repeated eigenvalue,
// a vanishing z_j, an iteration that does not settle, a result that fails the diagonal identity
// XtX'_ii = sum_k mu_k q_ik^2 -- and the model is decomposed afresh.

#define FOKL_POOL_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

// sum_j z2_j / d_j and sum_j z2_j / d_j^2 over [lo, hi) with d_j = (lam_j - pole) - t, stored to delta.  Eight partial sums
// each, added up in a fixed order: the same arithmetic (IEEE division, no contraction) whatever the vector width of the clone.
FOKL_POOL_CLONES void secular_sums(const double *lam, const double *z2, int lo, int hi, double pole, double t, double *delta,
                                   double *sum, double *dsum)
{
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = lo;
    for (; j + 8 <= hi; j += 8)
        for (int l = 0; l < 8; ++l) {
            const double d = (lam[j + l] - pole) - t;
            const double inv = 1.0 / d, term = z2[j + l] * inv;
            delta[j + l] = d;
            s[l] += term;
            ds[l] += term * inv;
        }
    for (int l = 0; j < hi; ++j, ++l) {
        const double d = (lam[j] - pole) - t;
        const double inv = 1.0 / d, term = z2[j] * inv;
        delta[j] = d;
        s[l] += term;
        ds[l] += term * inv;
    }
    *sum = ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7]));
    *dsum = ((ds[0] + ds[4]) + (ds[2] + ds[6])) + ((ds[1] + ds[5]) + (ds[3] + ds[7]));
}

// z^_j^2 as products of ratios in (0, inf) (interlacing): (mu_k - lam_j) / (lam_k - lam_j) for k < j, / (lam_k+1 - lam_j) for
// k >= j; D[k * n + j] = lam_j - mu_k.
FOKL_POOL_CLONES void secular_zhat(const double *lam, const double *D, int n, int m, double *zh)
{
    for (int j = 0; j < n; ++j) zh[j] = 1.0;
    for (int k = 0; k < m; ++k) {
        const double *d = D + (size_t)k * n;
        const double above = lam[k + 1], below = lam[k];
        for (int j = 0; j <= k; ++j) zh[j] *= -d[j] / (above - lam[j]);
        for (int j = k + 1; j < n; ++j) zh[j] *= -d[j] / (below - lam[j]);
    }
}

// x_k[j] = z^_j / (lam_j - mu_k) in place of the differences, each x_k normalised; false if one cannot be.
FOKL_POOL_CLONES bool secular_vectors(const double *zh, double *D, int n, int m)
{
    for (int k = 0; k < m; ++k) {
        double *d = D + (size_t)k * n;
        double ss[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int j = 0;
        for (; j + 8 <= n; j += 8)
            for (int l = 0; l < 8; ++l) {
                d[j + l] = zh[j + l] / d[j + l];
                ss[l] += d[j + l] * d[j + l];
            }
        for (int l = 0; j < n; ++j, ++l) {
            d[j] = zh[j] / d[j];
            ss[l] += d[j] * d[j];
        }
        const double total = ((ss[0] + ss[4]) + (ss[2] + ss[6])) + ((ss[1] + ss[5]) + (ss[3] + ss[7]));
        if (!(total > 0.0) || !std::isfinite(total)) return false;
        const double r = 1.0 / std::sqrt(total);
        for (int i = 0; i < n; ++i) d[i] *= r;
    }
    return true;
}

// One root: k-th interval.  delta[j] <- lam_j - mu_k.  Returns false when the iteration does not settle.
bool secular_root(const double *lam, const double *z2, int n, int k, double *delta, double *mu)
{
    const double eps = 2.220446049250313e-16;
    const double gap = lam[k + 1] - lam[k], half = 0.5 * gap;
    // which half of the interval: the sign of g at its middle (g rises from -inf to +inf across the interval) -- which also
    // is the first iterate
    double psi, dpsi, phi, dphi;
    secular_sums(lam, z2, 0, k + 1, lam[k], half, delta, &psi, &dpsi);       // every term negative
    secular_sums(lam, z2, k + 1, n, lam[k], half, delta, &phi, &dphi);       // every term positive
    const int o = psi + phi > 0.0 ? k : k + 1;
    const double pole = lam[o];
    double lo = o == k ? 0.0 : -half, hi = o == k ? half : 0.0;
    double t = o == k ? half : -half;
    const double dk = lam[k] - pole, dk1 = lam[k + 1] - pole;
    bool settled = false;
    for (int it = 0; it < 80; ++it) {
        const double g = psi + phi, mag = phi - psi;
        if (!(std::fabs(g) > (double)n * eps * mag)) {      // also leaves on NaN
            settled = g == g;
            break;
        }
        if (g > 0.0)
            hi = t;
        else
            lo = t;
        // psi ~ s1 + a1 / (dk - t), phi ~ s2 + a2 / (dk1 - t): value and slope at t (the two nearest poles exactly)
        const double e1 = dk - t, e2 = dk1 - t;
        const double a1 = dpsi * e1 * e1, a2 = dphi * e2 * e2;
        const double sr = (psi - dpsi * e1) + (phi - dphi * e2);
        // sr (dk - t)(dk1 - t) + a1 (dk1 - t) + a2 (dk - t) = 0
        const double qa = sr, qb = -(sr * (dk + dk1) + a1 + a2), qc = sr * dk * dk1 + a1 * dk1 + a2 * dk;
        double next = 0.5 * (lo + hi);
        if (qa == 0.0) {
            if (qb != 0.0) {
                const double r = -qc / qb;
                if (r > lo && r < hi) next = r;
            }
        } else {
            const double disc = qb * qb - 4.0 * qa * qc;
            if (disc >= 0.0) {
                const double sq = std::sqrt(disc);
                const double q = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
                const double r1 = q / qa, r2 = q != 0.0 ? qc / q : r1;
                if (r1 > lo && r1 < hi)
                    next = r1;
                else if (r2 > lo && r2 < hi)
                    next = r2;
            }
        }
        const bool stuck = next == t || !(hi - lo > 0.0);    // the bracket is down to neighbouring numbers
        if (!stuck) t = next;
        // the differences from the pole that was chosen (the first pass measured them from lam_k)
        secular_sums(lam, z2, 0, k + 1, pole, t, delta, &psi, &dpsi);
        secular_sums(lam, z2, k + 1, n, pole, t, delta, &phi, &dphi);
        if (stuck) {
            settled = true;
            break;
        }
    }
    if (!settled) return false;
    if (delta[k] >= 0.0 || delta[k + 1] <= 0.0) return false;   // mu_k must lie strictly inside its interval
    *mu = pole + t;
    return true;
}

// -> FOKL_OK with *used = 1 (job->lamb_out / qt_out hold the eigenpairs) or *used = 0 (the caller decomposes afresh)
int spectral_from_parent(fokl_host_pool *pool, fokl_host_job *job, bool *used)
{
    *used = false;
    const int m = (int)job->idx.size(), n = m + 1, c = job->parent_pos;
    const double *lam = job->parent_lamb, *P = job->parent_qt;     // P[j * n + i]: component i of eigenvector j
    if (!pool->dgemm || pool->lapack_signs || !lam || !P || c < 0 || c >= n || m < 1) return FOKL_OK;
    static thread_local std::vector<double> z, z2, D, zh;
    if (z.size() < (size_t)n) z.resize((size_t)n), z2.resize((size_t)n), zh.resize((size_t)n);
    if (D.size() < (size_t)n * m) D.resize((size_t)n * m);
    const double eps = 2.220446049250313e-16;
    const double scale = std::max(std::fabs(lam[0]), std::fabs(lam[n - 1]));
    double norm = 0.0;
    for (int j = 0; j < n; ++j) {
        z[j] = P[(size_t)j * n + c];
        z2[j] = z[j] * z[j];
        norm += z2[j];
        if (!(std::fabs(z[j]) > 1e-12)) return FOKL_OK;            // lam_j (nearly) stays an eigenvalue: deflation, not done here
    }
    if (!(std::fabs(norm - 1.0) < 1e-10)) return FOKL_OK;
    if (!(lam[0] > pool->singular * lam[n - 1])) return FOKL_OK;             // numerically singular: dsyevr decides (see `singular`)
    for (int j = 0; j + 1 < n; ++j)
        if (!(lam[j + 1] - lam[j] > 1024.0 * eps * scale)) return FOKL_OK;   // (nearly) repeated eigenvalue
    // roots; D[k * n + j] = lam_j - mu_k
    double *mu = job->lamb_out;
    for (int k = 0; k < m; ++k)
        if (!secular_root(lam, z2.data(), n, k, D.data() + (size_t)k * n, mu + k)) return FOKL_OK;
    secular_zhat(lam, D.data(), n, m, zh.data());
    for (int j = 0; j < n; ++j) {
        if (!(zh[j] > 0.0) || !(zh[j] < 4.0)) return FOKL_OK;      // (z^ is a row of an orthogonal matrix as z is)
        zh[j] = z[j] < 0.0 ? -std::sqrt(zh[j]) : std::sqrt(zh[j]);
    }
    if (!secular_vectors(zh.data(), D.data(), n, m)) return FOKL_OK;
    // eigenvectors: (rows != c of Q) X.  In Fortran's column-major reading the parent's Qt IS Q (n x n), D is X (n x m) and
    // qt_out is the result (m x m, column k = eigenvector k): two calls, the rows above and below c
    char nn = 'N';
    double one = 1.0, zero = 0.0;
    int nfull = n, mm = m;
    if (c > 0) {
        int rows = c;
        pool->dgemm(&nn, &nn, &rows, &mm, &nfull, &one, const_cast<double *>(P), &nfull, D.data(), &nfull, &zero, job->qt_out, &mm);
    }
    if (c < n - 1) {
        int rows = n - 1 - c;
        pool->dgemm(&nn, &nn, &rows, &mm, &nfull, &one, const_cast<double *>(P) + c + 1, &nfull, D.data(), &nfull, &zero,
                    job->qt_out + c, &mm);
    }
    // the diagonal of XtX' from the eigenpairs against the Gram itself: catches a parent that is not this model's (another
    // column order) and any loss of accuracy above
    const double *g = job->gram;
    const size_t ld = (size_t)job->ld;
    const double *Q = job->qt_out;
    double worst = 0.0, largest = 0.0;
    for (int i = 0; i < m; ++i) zh[i] = 0.0;
    for (int k = 0; k < m; ++k) {
        const double *v = Q + (size_t)k * m;
        for (int i = 0; i < m; ++i) zh[i] += mu[k] * v[i] * v[i];
    }
    for (int i = 0; i < m; ++i) {
        const double aii = g[(size_t)job->idx[i] * ld + job->idx[i]];
        worst = std::max(worst, std::fabs(aii - zh[i]));
        largest = std::max(largest, std::fabs(aii));
    }
    if (!(worst <= 1e-11 * largest)) return FOKL_OK;
    *used = true;
    return FOKL_OK;
}

int spectral_update(fokl_host_pool *pool, fokl_host_job *job, std::string &err)
{
    bool used = false;
    if (!job->parent_failed) {
        const int rc = spectral_from_parent(pool, job, &used);
        if (rc != FOKL_OK) return rc;
    }
    if (job->updated_out) *job->updated_out = used ? 1 : 0;
    if (!used) return spectral(pool, job, err);
    const int n = (int)job->idx.size();
    static thread_local std::vector<double> xty;
    if (xty.size() < (size_t)n) xty.resize((size_t)n);
    for (int i = 0; i < n; ++i) xty[i] = job->gram[(size_t)job->idx[i] * (size_t)job->ld + job->ycol];
    spectral_tail(pool, job, xty.data());
    return FOKL_OK;
}

void settle(fokl_host_job *job);

// A finish thread's share of one tape: blocks part, part + parts, ... of `block` rows, each as soon as the walk has
// passed it -- rows -> numbers (fokl_stream_expand), then, for tapes a host chain reads, the normals completed in place;
// block_done[blk] = 1 (release) after each, -1 for every block still open when the tape is sent back.
int expand_tape_blocks(fokl_host_pool *pool, fokl_host_job *job)
{
    const int nblocks = (job->draws + job->block - 1) / job->block;
    int32_t ready = 0;
    if (fokl_host_job *tape = job->parent) {                // asleep until the walker gets to the tape
        for (int spins = 0, lim = fokl_spin_budget(200); spins < lim && !tape->started.load(std::memory_order_acquire); ++spins) _mm_pause();
        if (!tape->started.load(std::memory_order_acquire)) {
            std::unique_lock<std::mutex> lock(pool->start_m);
            pool->start_cv.wait(lock, [&] { return tape->started.load(std::memory_order_acquire) != 0; });
        }
    }
    for (int blk = job->part; blk < nblocks; blk += job->parts) {
        const int k0 = blk * job->block, k1 = std::min(job->draws, k0 + job->block);
        int rc = FOKL_OK;
        for (int spins = 0; ready < k1;) {
            ready = __atomic_load_n(job->progress, __ATOMIC_ACQUIRE);
            if (ready < 0) break;
            if (ready < k1) {
                if (++spins < fokl_spin_budget(2000))
                    _mm_pause();
                else
                    std::this_thread::sleep_for(std::chrono::microseconds(10));
            }
        }
        if (ready >= 0) {
            rc = fokl_stream_expand(pool->stream, job->p1, job->astar, job->atau_star, job->rows, k0, k1, job->normals,
                                    job->pair_r2, job->lead, job->gam_sig, job->gam_tau);
            if (rc == FOKL_OK && job->finish_normals)
                fokl_finish_tape_rows(job->p1, job->normals, job->pair_r2, job->lead, k0, k1);
        }
        if (ready < 0 || rc != FOKL_OK) {
            for (int later = blk; later < nblocks; later += job->parts)
                __atomic_store_n(job->block_done + later, -1, __ATOMIC_RELEASE);
            return ready < 0 ? FOKL_ERR_STATE : rc;
        }
        __atomic_store_n(job->block_done + blk, 1, __ATOMIC_RELEASE);
    }
    return FOKL_OK;
}

// The eigenpairs of `job` are there (or never will be): the first of the jobs that derive theirs from them is returned -- the
// calling thread runs it next, the parent's vectors still in its cache, no hand-over through the queue -- and the others go
// to the FRONT of the queue.
fokl_host_job *release_dependents(fokl_host_pool *pool, fokl_host_job *job, int rc)
{
    std::vector<fokl_host_job *> deps;
    {
        std::lock_guard<std::mutex> lk(pool->dep_m);
        job->has_run = true;
        job->run_status = rc;
        deps.swap(job->dependents);
    }
    if (deps.empty()) return nullptr;
    for (fokl_host_job *d : deps) d->parent_failed = rc != FOKL_OK;
    if (deps.size() > 1) {
        {
            std::lock_guard<std::mutex> lk(pool->spectral_q.m);
            for (auto it = deps.rbegin(); it + 1 != deps.rend(); ++it) pool->spectral_q.q.push_front(*it);
        }
        if (deps.size() > 2)
            pool->spectral_q.cv.notify_all();
        else
            pool->spectral_q.cv.notify_one();
    }
    return deps.front();
}

// -> a job to run next on this thread (a spectral job's first dependent), or NULL
fokl_host_job *run(fokl_host_pool *pool, fokl_host_job *job)
{
    const auto t0 = std::chrono::steady_clock::now();
    fokl_host_job *next = nullptr;
    int rc = FOKL_OK;
    std::string err;
    std::atomic<int64_t> *busy = nullptr;
    switch (job->kind) {
    case Kind::noise:
        break;                                              // the noise thread has its own loop (noise_worker)
    case Kind::finish:
        rc = expand_tape_blocks(pool, job);
        busy = &pool->finish_busy_ns;
        break;
    case Kind::chain:
        if (job->block_done) {
            rc = fokl_gibbs_chain_from_finished_tape(job->lamb, job->qty, job->p1, job->b, job->btau, job->dtd,
                                                     job->sigsqd0, job->tausqd0, job->draws, job->normals,
                                                     job->gam_sig, job->gam_tau, job->block_done, job->block,
                                                     job->w_out, nullptr, nullptr, job->bstar_negative);
            if (rc != FOKL_OK) err = "chain: invalid arguments or the tape producer failed";
            if (rc == FOKL_OK && job->then) job->then(job->then_arg);       // (the draws are complete: what follows from them)
            busy = &pool->chain_busy_ns;
            break;
        }
        rc = fokl_gibbs_chain_from_raw_blocks(job->lamb, job->qty, job->p1, job->b, job->btau, job->dtd, job->sigsqd0,
                                              job->tausqd0, job->draws, job->normals, job->pair_r2, job->lead,
                                              job->gam_sig, job->gam_tau, job->raw_block_done, job->block, job->w_out,
                                              job->bstar_negative);
        if (rc != FOKL_OK) err = "chain: invalid arguments or the tape producer failed";
        if (rc == FOKL_OK && job->then) job->then(job->then_arg);
        busy = &pool->chain_busy_ns;
        break;
    case Kind::spectral:
        rc = job->parent_lamb ? spectral_update(pool, job, err) : spectral(pool, job, err);
        next = release_dependents(pool, job, rc);
        busy = &pool->spectral_busy_ns;
        break;
    }
    if (busy) {
        const auto dt = std::chrono::steady_clock::now() - t0;
        busy->fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(dt).count(), std::memory_order_relaxed);
    }
    if (job->self_owned) {
        if (job->parent) settle(job->parent);
        delete job;                                         // failures reach the chain job through block_done
        return next;
    }
    finish(job, rc, err.c_str());
    return next;
}

void worker(fokl_host_pool *pool, Queue *queue)
{
    ThreadCpuNote note{queue == &pool->chain_q ? 1 : queue == &pool->spectral_q ? 3 : 2};
    prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0);               // the short sleeps of the tape followers mean what they say
    for (;;) {
        fokl_host_job *job;
        {
            std::unique_lock<std::mutex> lock(queue->m);
            queue->cv.wait(lock, [&] { return queue->stop || !queue->q.empty(); });
            if (queue->q.empty()) return;                   // stop requested and the queue is drained
            job = queue->q.front();
            queue->q.pop_front();
        }
        while (job) job = run(pool, job);
    }
}

// The noise thread.  Tapes are recorded strictly in submission order.  Tentative tapes -- recorded ahead of the decision
// that they are needed -- may be NESTED: up to kMaxSpeculation of them can be on record without a verdict, each with
// the state of the stream at its beginning.  Verdicts settle from both ends of that list: a commit of the oldest makes
// it final; an abort of the oldest puts the stream back where it began and takes every younger one with it (the caller
// resolves those to "abort" as well: their content no longer is what the stream serves there); an abort of the youngest
// rewinds just that one.  A plain (non-tentative) request waits until nothing speculative is left.
struct Speculation {
    fokl_host_job *job;
    fokl_stream_cursor at_start;            // a rewind moves the walker; the stream's bulk data stays where it is
};

constexpr size_t kMaxSpeculation = 64;

void trace_noise(fokl_host_pool *pool, fokl_host_job *job, int64_t verdict_seen)
{
    if (!pool->trace_path.empty())
        pool->trace.push_back({job->t_submit, job->t_start, job->t_recorded, verdict_seen, job->p1,
                               job->tentative ? 1 : 0, job->verdict.load(std::memory_order_acquire)});
}

void settle(fokl_host_job *job)                            // status / error were set when the tape was recorded
{
    if (job->pending.fetch_sub(1, std::memory_order_acq_rel) != 1) return;      // somebody is still on the tape
    fokl_host_pool *pool = job->pool;                       // the job may be freed by its waiter right after `done`
    if (job->held && !job->span_out) {                      // nobody reads this tape's part of the stream any more
        job->held = false;
        fokl_stream_release(pool->stream, job->hold);
    }
    {
        std::lock_guard<std::mutex> lock(pool->done_m);
        job->done.store(1, std::memory_order_release);
    }
    pool->done_cv.notify_all();
}

void mark_started(fokl_host_pool *pool, fokl_host_job *job)
{
    if (job->pending.load(std::memory_order_relaxed) > 1) {             // finish jobs were submitted with it
        {
            std::lock_guard<std::mutex> lock(pool->start_m);
            job->started.store(1, std::memory_order_release);
        }
        pool->start_cv.notify_all();
    } else {
        job->started.store(1, std::memory_order_release);
    }
}

void record_tape(fokl_host_pool *pool, fokl_host_job *job)
{
    mark_started(pool, job);
    const auto t0 = std::chrono::steady_clock::now();
    job->t_start = std::chrono::duration_cast<std::chrono::nanoseconds>(t0.time_since_epoch()).count();
    int rc = fokl_stream_hold(pool->stream, &job->hold);
    job->held = rc == FOKL_OK;
    if (job->span_out) job->span_out[0] = job->hold;
    if (rc == FOKL_OK && job->rows_only) {
        // the walk writes progress itself; the span must be there before the last block is published
        int32_t local = 0;
        rc = fokl_stream_walk(pool->stream, job->p1, job->draws, job->astar, job->atau_star, job->rows, job->gam_sig,
                              job->gam_tau, &local);
        if (rc == FOKL_OK) {
            fokl_stream_cursor at;
            fokl_stream_tell(pool->stream, &at);
            if (job->span_out) job->span_out[1] = at.position;
        }
        __atomic_store_n(job->progress, rc == FOKL_OK ? job->draws : -1, __ATOMIC_RELEASE);
    } else if (rc == FOKL_OK && !pool->finish_q.empty()) {
        // the finish threads follow `progress` and turn the rows into numbers
        rc = fokl_stream_walk(pool->stream, job->p1, job->draws, job->astar, job->atau_star, job->rows, job->gam_sig,
                              job->gam_tau, job->progress);
    } else if (rc == FOKL_OK) {
        // no finish threads in this pool: this thread materialises each block itself
        const int block = job->block > 0 ? job->block : FOKL_TAPE_BLOCK;
        for (int k0 = 0; k0 < job->draws && rc == FOKL_OK; k0 += block) {
            const int k1 = std::min(job->draws, k0 + block);
            rc = fokl_stream_walk(pool->stream, job->p1, k1 - k0, job->astar, job->atau_star, job->rows + k0,
                                  job->gam_sig + k0, job->gam_tau + k0, nullptr);
            if (rc == FOKL_OK)
                rc = fokl_stream_expand(pool->stream, job->p1, job->astar, job->atau_star, job->rows, k0, k1, job->normals,
                                        job->pair_r2, job->lead, job->gam_sig, job->gam_tau);
            if (rc != FOKL_OK) break;
            if (job->block_done) __atomic_store_n(job->block_done + k0 / block, 1, __ATOMIC_RELEASE);
            __atomic_store_n(job->progress, k1, __ATOMIC_RELEASE);
        }
        if (rc != FOKL_OK) {
            if (job->block_done)
                for (int blk = 0; blk < (job->draws + block - 1) / block; ++blk) {
                    int32_t open = 0;
                    __atomic_compare_exchange_n(job->block_done + blk, &open, -1, false, __ATOMIC_ACQ_REL,
                                                __ATOMIC_ACQUIRE);
                }
            __atomic_store_n(job->progress, -1, __ATOMIC_RELEASE);
        }
    }
    if (rc != FOKL_OK) {
        job->status = rc;
        job->error = "noise tape: invalid arguments, gamma shape or the stream's producers failed";
    }
    fokl_stream_advance_floor(pool->stream);                // what is still needed behind the walker is held by its tape
    const auto t1 = std::chrono::steady_clock::now();
    job->t_recorded = std::chrono::duration_cast<std::chrono::nanoseconds>(t1.time_since_epoch()).count();
    pool->noise_busy_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count(),
                                  std::memory_order_relaxed);
}

void noise_worker(fokl_host_pool *pool)
{
    ThreadCpuNote note{0};
    prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0);
    Queue *queue = &pool->noise_q;
    std::deque<Speculation> open;                           // recorded, no verdict yet; oldest first
    auto aborted = [&](fokl_host_job *job) {
        __atomic_store_n(job->progress, -1, __ATOMIC_RELEASE);          // nobody may follow this tape
        // a chain started ahead follows the tape's block flags: with finish jobs on the tape they pass the -1 on (woken
        // below); without, this thread materialises the blocks itself and has to say so itself -- a tape sent back before
        // it was begun used to leave such a chain waiting for ever
        if (job->block_done && job->pending.load(std::memory_order_acquire) <= 1) {
            const int block = job->block > 0 ? job->block : FOKL_TAPE_BLOCK;
            for (int blk = 0; blk < (job->draws + block - 1) / block; ++blk) {
                int32_t open = 0;
                __atomic_compare_exchange_n(job->block_done + blk, &open, -1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
            }
        }
        mark_started(pool, job);                                        // (finish threads asleep on it see the -1)
        trace_noise(pool, job, now_ns());
        settle(job);
    };
    auto first_aborted = [&]() -> size_t {
        size_t i = 0;
        while (i < open.size() && open[i].job->verdict.load(std::memory_order_acquire) >= 0) ++i;
        return i;
    };
    auto settle_verdicts = [&] {
        while (!open.empty() && open.front().job->verdict.load(std::memory_order_acquire) > 0) {
            fokl_host_job *job = open.front().job;
            open.pop_front();
            trace_noise(pool, job, now_ns());
            settle(job);
        }
        const size_t cut = first_aborted();
        if (cut < open.size()) {
            fokl_stream_seek(pool->stream, &open[cut].at_start);
            for (size_t k = cut; k < open.size(); ++k) aborted(open[k].job);
            open.erase(open.begin() + (std::ptrdiff_t)cut, open.end());
        }
    };
    for (;;) {
        // verdicts that can be acted on: commits from the old end; an abort ANYWHERE takes that tape and every younger
        // one with it (the caller resolves those to "abort" as well -- it sends its orders back youngest first) and puts
        // the stream back where the oldest aborted tape began
        settle_verdicts();
        if (pool->test_delay_us > 0)                        // tests: widen the window between the verdicts and the queue
            std::this_thread::sleep_for(std::chrono::microseconds(pool->test_delay_us));
        // the next request, if it can be served now
        fokl_host_job *job = nullptr;
        bool stopping = false, must_settle = false;
        {
            std::unique_lock<std::mutex> lock(queue->m);
            if (open.empty()) {
                const auto w0 = std::chrono::steady_clock::now();
                queue->cv.wait(lock, [&] { return queue->stop || !queue->q.empty(); });
                pool->noise_queue_wait_ns.fetch_add(
                    std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count(),
                    std::memory_order_relaxed);
                if (queue->q.empty()) return;               // stop requested and the queue is drained
            }
            stopping = queue->stop;
            if (!queue->q.empty()) {
                fokl_host_job *next = queue->q.front();
                // A request that was queued AFTER the driver sent tapes back must not be recorded behind those tapes
                // (its content would be what the stream serves after them, not after the rewind).  Taking the queue's
                // mutex makes the verdicts stored before that submit visible here: look again before recording
                // anything behind what is open, and let the verdict step rewind first.
                if (!open.empty() && first_aborted() < open.size())
                    must_settle = true;
                else if (open.empty() || (next->tentative && open.size() < kMaxSpeculation)) {
                    job = next;
                    queue->q.pop_front();
                }
            }
        }
        if (must_settle) continue;
        if (!job) {
            // something speculative is open and nothing can be recorded behind it: its verdict is what comes next
            (void)stopping;
            const auto w0 = std::chrono::steady_clock::now();
            auto settled = [&] {
                return open.front().job->verdict.load(std::memory_order_acquire) > 0 || first_aborted() < open.size();
            };
            auto more = [&] {
                if (open.size() >= kMaxSpeculation) return false;
                std::lock_guard<std::mutex> lock(queue->m);
                return !queue->q.empty() && queue->q.front()->tentative;
            };
            for (int spins = 0; !settled(); ++spins) {
                if (spins < fokl_spin_budget(4000)) {
                    _mm_pause();
                    if ((spins & 63) == 63 && more()) break;
                } else {
                    // asleep until a verdict (fokl_pool_resolve) or a request (submit) wakes the queue's condition; the
                    // timeout only covers a verdict stored between the test above and the wait
                    std::unique_lock<std::mutex> lock(queue->m);
                    if (settled()) break;
                    if (open.size() < kMaxSpeculation && !queue->q.empty() && queue->q.front()->tentative) break;
                    // (system clock: libstdc++ then waits with pthread_cond_timedwait, which ThreadSanitizer follows; the
                    // steady clock's pthread_cond_clockwait it does not, and reports the mutex as held across the wait)
                    queue->cv.wait_until(lock, std::chrono::system_clock::now() + std::chrono::microseconds(200));
                    if (open.size() < kMaxSpeculation && !queue->q.empty() && queue->q.front()->tentative) break;
                }
            }
            pool->noise_verdict_wait_ns.fetch_add(
                std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count(),
                std::memory_order_relaxed);
            continue;
        }
        if (job->tentative) {
            if (job->verdict.load(std::memory_order_acquire) < 0) {     // aborted before it was started
                aborted(job);
                continue;
            }
            open.push_back({job, {}});
            fokl_stream_tell(pool->stream, &open.back().at_start);
            record_tape(pool, job);
        } else {
            record_tape(pool, job);
            trace_noise(pool, job, job->t_recorded);
            settle(job);
        }
    }
}

void submit(Queue &queue, fokl_host_job *job)
{
    {
        std::lock_guard<std::mutex> lock(queue.m);
        queue.q.push_back(job);
    }
    queue.cv.notify_one();
}

void stop(Queue &queue)
{
    {
        std::lock_guard<std::mutex> lock(queue.m);
        queue.stop = true;
    }
    queue.cv.notify_all();
}

}  // namespace

extern "C" int fokl_pool_create(int chain_threads, int finish_threads, int spectral_threads, int bulk_threads,
                                int noise_cpu, void *dsyevr, uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss,
                                double *gauss_cache, uint32_t *prestate_ring, int prestate_entries, fokl_host_pool **out)
{
    if (!out || chain_threads < 1 || bulk_threads < 1 || bulk_threads > 16 || spectral_threads < 0 || chain_threads > 64 || spectral_threads > 64 ||
        finish_threads < 0 || finish_threads > 64 || !mt_key ||
        !mt_pos || !has_gauss || !gauss_cache || (spectral_threads > 0 && !dsyevr)) {
        fokl_set_global_error("fokl_pool_create: bad thread counts, null RNG state or missing dsyevr");
        return FOKL_ERR_ARG;
    }
    if (*mt_pos < 0 || *mt_pos > 624) {
        fokl_set_global_error("fokl_pool_create: invalid MT19937 position");
        return FOKL_ERR_ARG;
    }
    auto *pool = new fokl_host_pool();
    if (fokl_stream_create(mt_key, *mt_pos, *has_gauss, *gauss_cache, bulk_threads, prestate_ring, prestate_entries,
                           &pool->stream) != FOKL_OK) {
        delete pool;
        return FOKL_ERR_STATE;
    }
    pool->dsyevr = reinterpret_cast<dsyevr_fn>(dsyevr);
    pool->mt_key = mt_key;
    pool->mt_pos = mt_pos;
    pool->has_gauss = has_gauss;
    pool->gauss_cache = gauss_cache;
    if (const char *path = std::getenv("FOKL_POOL_TRACE")) pool->trace_path = path;
    if (const char *us = std::getenv("FOKL_POOL_TEST_DELAY_US")) pool->test_delay_us = std::max(0, std::atoi(us));
    if (const char *signs = std::getenv("FOKL_EIGH_SIGNS")) pool->lapack_signs = std::strcmp(signs, "lapack") == 0;
    if (const char *sing = std::getenv("FOKL_EIGH_SINGULAR")) pool->singular = std::max(0.0, std::atof(sing));
    try {
        pool->threads.emplace_back(noise_worker, pool);
        if (noise_cpu >= 0 && noise_cpu < CPU_SETSIZE) {
            // the random stream is the serial resource of a fit: its thread gets a logical CPU of its own (the caller
            // keeps every other thread of the process off that core); failure to pin is not an error
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(noise_cpu, &set);
            pthread_setaffinity_np(pool->threads.back().native_handle(), sizeof(set), &set);
        }
        for (int i = 0; i < chain_threads; ++i) pool->threads.emplace_back(worker, pool, &pool->chain_q);
        pool->finish_q.resize((size_t)finish_threads);
        for (auto &q : pool->finish_q) pool->threads.emplace_back(worker, pool, &q);
        for (int i = 0; i < spectral_threads; ++i) {
            pool->spectral_thread_ids.push_back(pool->threads.size());
            pool->threads.emplace_back(worker, pool, &pool->spectral_q);
        }
    } catch (const std::exception &e) {
        stop(pool->noise_q);
        stop(pool->chain_q);
        stop(pool->spectral_q);
        for (auto &q : pool->finish_q) stop(q);
        for (auto &t : pool->threads) t.join();
        fokl_stream_destroy(pool->stream);
        delete pool;
        fokl_set_global_error(std::string("fokl_pool_create: ") + e.what());
        return FOKL_ERR_STATE;
    }
    *out = pool;
    return FOKL_OK;
}

// Finishes everything that was submitted (every recorded tape advances the stream, used or not), then stops.
extern "C" void fokl_pool_destroy(fokl_host_pool *pool)
{
    if (!pool) return;
    stop(pool->noise_q);
    stop(pool->chain_q);
    stop(pool->spectral_q);
    for (auto &q : pool->finish_q) stop(q);
    for (auto &t : pool->threads) t.join();
    // the stream goes back to its owner where the walker stands: numpy's state tuple after everything that was walked
    fokl_stream_state(pool->stream, pool->mt_key, pool->mt_pos, pool->has_gauss, pool->gauss_cache);
    fokl_stream_destroy(pool->stream);
    if (!pool->trace_path.empty()) {
        if (FILE *f = std::fopen(pool->trace_path.c_str(), "a")) {
            for (const auto &r : pool->trace)
                std::fprintf(f, "noise %lld %lld %lld %lld %lld %lld %lld\n", (long long)r[0], (long long)r[1],
                             (long long)r[2], (long long)r[3], (long long)r[4], (long long)r[5], (long long)r[6]);
            std::fclose(f);
        }
    }
    delete pool;
}

static void submit_finish_jobs(fokl_host_pool *pool, fokl_host_job *parent, bool finish_normals)
{
    const int parts = (int)pool->finish_q.size();
    for (int part = 0; part < parts; ++part) {             // the tape is materialised by all finish threads
        auto *fin = new fokl_host_job();
        fin->pool = pool;
        fin->kind = Kind::finish;
        fin->self_owned = true;
        fin->parent = parent;
        fin->p1 = parent->p1;
        fin->draws = parent->draws;
        fin->astar = parent->astar;
        fin->atau_star = parent->atau_star;
        fin->rows = parent->rows;
        fin->normals = parent->normals;
        fin->pair_r2 = parent->pair_r2;
        fin->lead = parent->lead;
        fin->gam_sig = parent->gam_sig;
        fin->gam_tau = parent->gam_tau;
        fin->progress = parent->progress;
        fin->block_done = parent->block_done;
        fin->block = parent->block;
        fin->finish_normals = finish_normals;
        fin->part = part;
        fin->parts = parts;
        submit(pool->finish_q[(size_t)part], fin);
    }
}

extern "C" int fokl_pool_submit_noise(fokl_host_pool *pool, int p1, int draws, double astar, double atau_star,
                                      fokl_tape_row *rows, double *normals, double *pair_r2, int32_t *lead,
                                      double *gam_sig, double *gam_tau, int32_t *progress, int tentative,
                                      int32_t *block_done, int block, int finish, uint64_t *span_out,
                                      fokl_host_job **out)
{
    const bool rows_only = finish == 2;
    if (!pool || !out || p1 <= 0 || draws < 0 || !rows || !gam_sig || !gam_tau || !progress ||
        (!rows_only && (!normals || !pair_r2 || !lead || !block_done || block < 1)) || (rows_only && !span_out)) {
        fokl_set_global_error("fokl_pool_submit_noise: null pointer, empty model or bad block size");
        return FOKL_ERR_ARG;
    }
    auto *job = new fokl_host_job();
    job->pool = pool;
    job->kind = Kind::noise;
    job->p1 = p1;
    job->draws = draws;
    job->astar = astar;
    job->atau_star = atau_star;
    job->rows = rows;
    job->normals = normals;
    job->pair_r2 = pair_r2;
    job->lead = lead;
    job->gam_sig = gam_sig;
    job->gam_tau = gam_tau;
    job->progress = progress;
    job->block_done = block_done;
    job->block = block;
    job->tentative = tentative != 0;
    job->span_out = span_out;
    job->rows_only = rows_only;
    if (!pool->trace_path.empty()) job->t_submit = now_ns();
    *out = job;
    if (!rows_only && !pool->finish_q.empty()) {
        // the finish threads follow the walk whether or not a chain has been asked for yet (a tape walked ahead of the
        // decision that it is needed is complete when its chain comes); a tape that is sent back ends these jobs too
        job->pending.store(1 + (int)pool->finish_q.size(), std::memory_order_relaxed);
        submit_finish_jobs(pool, job, finish == 1);
    }
    submit(pool->noise_q, job);
    return FOKL_OK;
}

extern "C" int fokl_pool_submit_chain(fokl_host_pool *pool, const double *lamb, const double *qty, int p1, double b,
                                      double btau, double dtd, double sigsqd0, double tausqd0, int draws,
                                      const double *normals, const double *pair_r2, const int32_t *lead,
                                      const double *gam_sig, const double *gam_tau, const int32_t *progress,
                                      int32_t *block_done, int block, int finishing_requested, double *w_out,
                                      int32_t *bstar_negative, void (*then)(void *), void *then_arg, fokl_host_job **out)
{
    if (!pool || !out || !lamb || !qty || p1 <= 0 || draws < 0 || !normals || !pair_r2 || !lead || !gam_sig ||
        !gam_tau || !progress || !w_out || !bstar_negative || !block_done || block < 1) {
        fokl_set_global_error("fokl_pool_submit_chain: null pointer or empty model");
        return FOKL_ERR_ARG;
    }
    auto *job = new fokl_host_job();
    job->pool = pool;
    job->kind = Kind::chain;
    job->lamb = lamb;
    job->qty = qty;
    job->p1 = p1;
    job->b = b;
    job->btau = btau;
    job->dtd = dtd;
    job->sigsqd0 = sigsqd0;
    job->tausqd0 = tausqd0;
    job->draws = draws;
    job->normals = const_cast<double *>(normals);
    job->pair_r2 = const_cast<double *>(pair_r2);
    job->lead = const_cast<int32_t *>(lead);
    job->gam_sig = const_cast<double *>(gam_sig);
    job->gam_tau = const_cast<double *>(gam_tau);
    job->progress = const_cast<int32_t *>(progress);
    job->w_out = w_out;
    job->bstar_negative = bstar_negative;
    job->block = block;
    job->then = then;
    job->then_arg = then_arg;
    if (finishing_requested)
        job->block_done = block_done;                       // the finish threads complete the normals in place
    else
        job->raw_block_done = block_done;                   // ... or only expand them: the chain completes each row
    *out = job;
    submit(pool->chain_q, job);
    return FOKL_OK;
}

static int new_spectral_job(const char *who, fokl_host_pool *pool, const double *gram, int ld, const int32_t *idx, int p1,
                            int ycol, double *lamb_out, double *qt_out, double *qty_out, double *betahat_out,
                            double *moments_out, fokl_host_job **job_out)
{
    const std::string name(who);
    if (!pool || !job_out || !gram || !idx || p1 <= 0 || ld <= 0 || ycol < 0 || ycol >= ld || !lamb_out || !qt_out ||
        !qty_out || !betahat_out) {
        fokl_set_global_error(name + ": null pointer, empty model or y column out of range");
        return FOKL_ERR_ARG;
    }
    if (!pool->dsyevr) {
        fokl_set_global_error(name + ": the pool was created without spectral threads");
        return FOKL_ERR_STATE;
    }
    for (int i = 0; i < p1; ++i)
        if (idx[i] < 0 || idx[i] >= ld) {
            fokl_set_global_error(name + ": column index out of range");
            return FOKL_ERR_ARG;
        }
    auto *job = new fokl_host_job();
    job->pool = pool;
    job->kind = Kind::spectral;
    job->gram = gram;
    job->ld = ld;
    job->ycol = ycol;
    job->idx.assign(idx, idx + p1);
    job->lamb_out = lamb_out;
    job->qt_out = qt_out;
    job->qty_out = qty_out;
    job->betahat_out = betahat_out;
    job->moments_out = moments_out;
    *job_out = job;
    return FOKL_OK;
}

extern "C" int fokl_pool_submit_spectral(fokl_host_pool *pool, const double *gram, int ld, const int32_t *idx, int p1,
                                         int ycol, double *lamb_out, double *qt_out, double *qty_out,
                                         double *betahat_out, double *moments_out, fokl_host_job **out)
{
    const int rc = new_spectral_job("fokl_pool_submit_spectral", pool, gram, ld, idx, p1, ycol, lamb_out, qt_out, qty_out,
                                    betahat_out, moments_out, out);
    if (rc == FOKL_OK) submit(pool->spectral_q, *out);
    return rc;
}

// G2 of a model from the eigenpairs of the model with ONE MORE column (`parent_*`: n = p1 + 1 eigenvalues ascending, Qt as
// fokl_pool_submit_spectral writes it; `parent_pos`: which of the parent's columns this model lacks).  parent_job: NULL
// when those arrays are complete, else the spectral job of this pool that writes them (not yet waited for): this job is
// queued when that one has run.  `updated` (may be NULL): whether the eigenpairs were derived from the parent's (1) or the
// model was decomposed afresh after all (0: the parent failed, repeated eigenvalues, a failed accuracy check, no dgemm
// bound, FOKL_EIGH_SIGNS=lapack).  Layout and sign convention of the results: the fresh job's.
extern "C" int fokl_pool_submit_spectral_update(fokl_host_pool *pool, const double *gram, int ld, const int32_t *idx, int p1,
                                                int ycol, const double *parent_lamb, const double *parent_qt,
                                                int parent_pos, fokl_host_job *parent_job, double *lamb_out,
                                                double *qt_out, double *qty_out, double *betahat_out, double *moments_out,
                                                int32_t *updated, fokl_host_job **out)
{
    if (!parent_lamb || !parent_qt || parent_pos < 0 || parent_pos > p1 ||
        (parent_job && (parent_job->kind != Kind::spectral || parent_job->pool != pool))) {
        fokl_set_global_error("fokl_pool_submit_spectral_update: no parent, position out of range or a job of another kind");
        return FOKL_ERR_ARG;
    }
    const int rc = new_spectral_job("fokl_pool_submit_spectral_update", pool, gram, ld, idx, p1, ycol, lamb_out, qt_out,
                                    qty_out, betahat_out, moments_out, out);
    if (rc != FOKL_OK) return rc;
    fokl_host_job *job = *out;
    job->parent_lamb = parent_lamb;
    job->parent_qt = parent_qt;
    job->parent_pos = parent_pos;
    job->updated_out = updated;
    if (parent_job) {
        std::lock_guard<std::mutex> lk(pool->dep_m);
        if (!parent_job->has_run) {
            parent_job->dependents.push_back(job);
            return FOKL_OK;
        }
        job->parent_failed = parent_job->run_status != FOKL_OK;
    }
    submit(pool->spectral_q, job);
    return FOKL_OK;
}

// BLAS dgemm (`fn`: its address, Fortran ABI with 32-bit integers -- scipy.linalg.cython_blas's) for the product of
// fokl_pool_submit_spectral_update; NULL: update jobs decompose afresh.  Before the first job.
extern "C" int fokl_pool_use_dgemm(fokl_host_pool *pool, void *fn)
{
    if (!pool) {
        fokl_set_global_error("fokl_pool_use_dgemm: null pool");
        return FOKL_ERR_ARG;
    }
    pool->dgemm = reinterpret_cast<dgemm_fn>(fn);
    return FOKL_OK;
}

// Is a tape that the pool records (and, with finish threads, expands block by block) complete?  1 = every row is recorded
// and every block is there, 0 = not yet, -1 = it never will be (the producer failed or the tape was sent back).  Acquire
// loads of the words the noise and finish threads publish with release stores: what a consumer on any thread may poll
// before it reads the tape's arrays.  progress / block_done may be NULL (not followed).
extern "C" int fokl_tape_ready(const int32_t *progress, int draws, const int32_t *block_done, int block)
{
    if (progress) {
        const int32_t p = __atomic_load_n(progress, __ATOMIC_ACQUIRE);
        if (p < 0) return -1;
        if (p < draws) return 0;
    }
    if (block_done && block > 0) {
        const int nblocks = (draws + block - 1) / block;
        for (int blk = nblocks - 1; blk >= 0; --blk) {
            const int32_t f = __atomic_load_n(block_done + blk, __ATOMIC_ACQUIRE);
            if (f < 0) return -1;
            if (f == 0) return 0;
        }
    }
    return 1;
}

// Verdict on a tentative noise job: commit != 0 keeps the tape (it is then exactly the tape a plain submission at that
// point of the stream would have recorded), 0 discards it and rewinds the stream to where the tape began.
extern "C" int fokl_pool_resolve(fokl_host_job *job, int commit)
{
    if (!job || job->kind != Kind::noise || !job->tentative) {
        fokl_set_global_error("fokl_pool_resolve: not a tentative noise job");
        return FOKL_ERR_ARG;
    }
    int expected = 0;
    if (!job->verdict.compare_exchange_strong(expected, commit ? 1 : -1)) {
        fokl_set_global_error("fokl_pool_resolve: the job has been resolved already");
        return FOKL_ERR_STATE;
    }
    job->pool->noise_q.cv.notify_one();                    // (a walker asleep on the verdicts of its open tapes)
    return FOKL_OK;
}

extern "C" int fokl_pool_poll(const fokl_host_job *job)
{
    return job && job->done.load(std::memory_order_acquire) ? 1 : 0;
}

// Blocks until the job has run, frees it and returns its status.
extern "C" int fokl_pool_wait(fokl_host_job *job)
{
    if (!job) {
        fokl_set_global_error("fokl_pool_wait: null job");
        return FOKL_ERR_ARG;
    }
    for (int spins = 0, lim = fokl_spin_budget(2000); spins < lim && !job->done.load(std::memory_order_acquire); ++spins) _mm_pause();
    if (!job->done.load(std::memory_order_acquire)) {
        fokl_host_pool *pool = job->pool;
        std::unique_lock<std::mutex> lock(pool->done_m);
        pool->done_cv.wait(lock, [&] { return job->done.load(std::memory_order_acquire) != 0; });
    }
    const int rc = job->status;
    if (rc != FOKL_OK) fokl_set_global_error("host pool job failed: " + job->error);
    delete job;
    return rc;
}

extern "C" int fokl_pool_busy_seconds(const fokl_host_pool *pool, double *noise, double *chain, double *finish,
                                      double *spectral)
{
    if (!pool) {
        fokl_set_global_error("fokl_pool_busy_seconds: null pool");
        return FOKL_ERR_ARG;
    }
    if (noise) *noise = 1e-9 * (double)pool->noise_busy_ns.load();
    if (chain) *chain = 1e-9 * (double)pool->chain_busy_ns.load();
    if (finish) *finish = 1e-9 * (double)pool->finish_busy_ns.load();
    if (spectral) *spectral = 1e-9 * (double)pool->spectral_busy_ns.load();
    return FOKL_OK;
}

// Models of `from_columns` columns or more are diagonalised by LAPACK's dsyevd (`fn`: its address, Fortran ABI with 32-bit
// integers -- scipy.linalg.cython_lapack's) instead of dsyevr; 0 / NULL: dsyevr for every size.  Before the first job.
extern "C" int fokl_pool_use_dsyevd(fokl_host_pool *pool, void *fn, int from_columns)
{
    if (!pool) {
        fokl_set_global_error("fokl_pool_use_dsyevd: null pool");
        return FOKL_ERR_ARG;
    }
    pool->dsyevd = reinterpret_cast<dsyevd_fn>(fn);
    pool->dsyevd_from = fn ? std::max(0, from_columns) : 0;
    return FOKL_OK;
}

// The spectral threads share no data with the threads around the random stream (bulk, noise, finish, chain): the caller may
// give them CPUs of their own -- another last-level cache domain -- instead of the affinity they inherited.
extern "C" int fokl_pool_spectral_affinity(fokl_host_pool *pool, const int32_t *cpus, int count)
{
    if (!pool || !cpus || count < 1) {
        fokl_set_global_error("fokl_pool_spectral_affinity: null pointer or empty CPU list");
        return FOKL_ERR_ARG;
    }
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int i = 0; i < count; ++i)
        if (cpus[i] >= 0 && cpus[i] < CPU_SETSIZE) CPU_SET(cpus[i], &set);
    int failed = 0;
    for (size_t id : pool->spectral_thread_ids)
        if (pthread_setaffinity_np(pool->threads[id].native_handle(), sizeof(set), &set) != 0) ++failed;
    if (failed) {
        fokl_set_global_error("fokl_pool_spectral_affinity: pthread_setaffinity_np failed");
        return FOKL_ERR_STATE;
    }
    return FOKL_OK;
}

extern "C" fokl_stream *fokl_pool_stream(fokl_host_pool *pool) { return pool ? pool->stream : nullptr; }

extern "C" int fokl_pool_release_hold(fokl_host_pool *pool, uint64_t position)
{
    if (!pool) {
        fokl_set_global_error("fokl_pool_release_hold: null pool");
        return FOKL_ERR_ARG;
    }
    return fokl_stream_release(pool->stream, position);
}

// The pool's random stream (fokl_stream_stats): CPU seconds of its bulk threads, seconds the walker waited for them,
// segments produced, gamma attempts walked and how many of them needed the exact expressions.
extern "C" int fokl_pool_stream_stats(const fokl_host_pool *pool, double *bulk_busy_s, double *walker_wait_s,
                                      int64_t *segments, int64_t *gamma_attempts, int64_t *gamma_attempts_exact)
{
    if (!pool) {
        fokl_set_global_error("fokl_pool_stream_stats: null pool");
        return FOKL_ERR_ARG;
    }
    return fokl_stream_stats(pool->stream, bulk_busy_s, walker_wait_s, segments, gamma_attempts, gamma_attempts_exact);
}

// Where the noise thread -- the serial resource of a fit -- was not recording: waiting for the next request with an
// empty queue, and holding the stream until the driver's verdict on a tentative tape.
extern "C" int fokl_pool_noise_waits(const fokl_host_pool *pool, double *queue_wait, double *verdict_wait)
{
    if (!pool) {
        fokl_set_global_error("fokl_pool_noise_waits: null pool");
        return FOKL_ERR_ARG;
    }
    if (queue_wait) *queue_wait = 1e-9 * (double)pool->noise_queue_wait_ns.load();
    if (verdict_wait) *verdict_wait = 1e-9 * (double)pool->noise_verdict_wait_ns.load();
    return FOKL_OK;
}
