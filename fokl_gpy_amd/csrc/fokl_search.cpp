// The sequential kill tests of one sub-stage, and what they share with the rest of a search, off the Python thread
// (round 4).  Restates engine.ForwardSelection._kill_tests_pipelined and the pieces of host_pipeline.py it drives -- the
// tapes on order, the eigen-decompositions submitted ahead, the chains on host threads or on the device, the decisions
// taken from guessed intercept scales and confirmed afterwards -- as native code next to the thread pool: the reference's
// loop FoKLRoutines.py:1666-1690, same tests, same order, same consumption of the random stream.  The Python driver took
// ~200 us per kill test (365 of them per benchmark fit: a 60-65 ms floor under the fit); one turn of this loop is a few
// microseconds of bookkeeping between waits for threads that are doing the work.
//
// What stays in Python (engine.ForwardSelection, a dozen calls per fit): the sequence of sub-stages, the K1 / K2 / K3
// launches, the statistics that order a sub-stage's proposals, the stop rule.  It shares with this file, through the
// fokl_search_* calls, the objects both sides handle:
//
//   tapes      one model evaluation's noise (rows walked by the pool's noise thread, materialised by its finish threads
//              into buffers from a size-class free list here -- page-locked when a device chain engine may read them);
//              the queue of tapes ON ORDER ahead of the decisions that they are needed (fokl_search_speculate /
//              tape_for: a tape is committed when the evaluation that comes has its size, else everything on order is
//              sent back and the walker rewound);
//   spectra    G2 jobs on the pool's spectral threads, result buffers owned here, reference counted;
//   outcomes   a model evaluation: its spectrum, BIC, tape and chain (host chain thread or device engine).
//
// Everything in a fokl_search is touched by ONE thread (the driver); the pool's and the device engine's own threads are
// reached through their C ABI only.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <immintrin.h>

#include "../../include/fokl_hip_internal.h"
#include "fokl_spin.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip

// fokl_sampler.cpp (library-internal): the chain of a tape whose blocks are materialised (not finished) by other threads
extern "C" __attribute__((visibility("hidden"))) int fokl_gibbs_chain_from_raw_blocks(
    const double *lamb, const double *qty, int p1, double b, double btau, double dtd, double sigsqd0, double tausqd0,
    int draws, const double *normals, const double *pair_r2, const int32_t *lead, const double *gam_sig,
    const double *gam_tau, const int32_t *block_done, int block, double *w_out, int32_t *bstar_negative);

namespace {

inline double now_s()
{
    return 1e-9 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
                      std::chrono::steady_clock::now().time_since_epoch())
                      .count();
}

// ---------------------------------------------------------------------------------------------------------------
// buffers: size classes of 64 K doubles, kept from fit to fit (fresh allocations would be page-faulted in by the pool's
// threads; page-locking costs milliseconds)
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t kClassDoubles = 65536;

struct BufferStore {
    std::mutex m;
    std::map<size_t, std::vector<double *>> free_plain, free_pinned;    // class -> blocks
    size_t kept = 0, limit = (size_t)2048 * 131072;                     // doubles kept (FOKL_HOST_POOL_MB)
    BufferStore()
    {
        if (const char *mb = std::getenv("FOKL_HOST_POOL_MB")) limit = (size_t)(std::atof(mb) * 131072.0);
    }
};

BufferStore &store()
{
    static BufferStore s;
    return s;
}

double *take_buffer(size_t doubles, bool pinned, size_t *classes, bool *got_pinned)
{
    const size_t cls = (doubles + kClassDoubles - 1) / kClassDoubles;
    *classes = cls;
    BufferStore &s = store();
    {
        std::lock_guard<std::mutex> lock(s.m);
        auto &lists = pinned ? s.free_pinned : s.free_plain;
        auto it = lists.find(cls);
        if (it != lists.end() && !it->second.empty()) {
            double *p = it->second.back();
            it->second.pop_back();
            s.kept -= cls * kClassDoubles;
            *got_pinned = pinned;
            return p;
        }
    }
    if (pinned) {
        void *p = nullptr;
        if (fokl_host_alloc(cls * kClassDoubles * sizeof(double), &p) == FOKL_OK && p) {
            *got_pinned = true;
            return static_cast<double *>(p);
        }
    }
    void *p = nullptr;
    if (posix_memalign(&p, 64, cls * kClassDoubles * sizeof(double)) != 0) return nullptr;
    *got_pinned = false;
    return static_cast<double *>(p);
}

void give_buffer(double *p, size_t classes, bool pinned)
{
    if (!p) return;
    BufferStore &s = store();
    {
        std::lock_guard<std::mutex> lock(s.m);
        if (s.kept + classes * kClassDoubles <= s.limit) {
            (pinned ? s.free_pinned : s.free_plain)[classes].push_back(p);
            s.kept += classes * kClassDoubles;
            return;
        }
    }
    if (pinned)
        (void)fokl_host_free(p);
    else
        std::free(p);
}

// ---------------------------------------------------------------------------------------------------------------
// the objects of a search
// ---------------------------------------------------------------------------------------------------------------

// live objects of each kind, process-wide (FOKL_SEARCH_PROFILE prints them when a search ends: a search that has been
// destroyed leaves none of its own behind)
struct Census {
    std::atomic<int64_t> tapes{0}, spectra{0}, outcomes{0};
};
Census &census()
{
    static Census c;
    return c;
}
template <typename T, std::atomic<int64_t> Census::*Member>
struct Counted {
    Counted() { (census().*Member).fetch_add(1, std::memory_order_relaxed); }
    Counted(const Counted &) { (census().*Member).fetch_add(1, std::memory_order_relaxed); }
    ~Counted() { (census().*Member).fetch_sub(1, std::memory_order_relaxed); }
};

struct Tape : Counted<Tape, &Census::tapes> {
    int p1 = 0, draws = 0;
    bool model = false;                     // ordered for a sub-stage's model (finished by host threads as it is walked)
    bool tentative = false, resolved = false, abandoned = false;
    double *mem = nullptr;
    size_t classes = 0;
    bool pinned = false;
    // carved out of mem (the layout of _capi.NoiseTape)
    double *normals = nullptr, *pair_r2 = nullptr, *gam_sig = nullptr, *gam_tau = nullptr;
    int32_t *progress = nullptr, *block_done = nullptr, *lead = nullptr;
    fokl_tape_row *rows = nullptr;
    bool finishing = false;                 // the finish threads complete the normals in place
    // a tape the DEVICE expands (fokl_dchain_submit_rows) exists as rows only: [progress | rows | gam_sig | gam_tau] in a
    // small page-locked block; materialise() gives it the arrays after all (a host chain has to read it)
    bool rows_only = false;
    double *rows_mem = nullptr;
    size_t rows_classes = 0;
    // [position the stream is held from, position behind the tape]; ~0: the walker never got to it (sent back before)
    uint64_t span[2] = {~(uint64_t)0, ~(uint64_t)0};
    bool holds_stream = false;
    double astar = 0, atau_star = 0;
    fokl_host_job *noise = nullptr;         // freed (fokl_pool_wait) when the tape goes
    int refs = 1;
    uint64_t seq = 0;                       // order of request = order in the walker's queue
    std::vector<fokl_host_job *> readers;   // host chains given up while they may still be reading the tape
};

struct Spectrum : Counted<Spectrum, &Census::spectra> {
    int p1 = 0;
    std::vector<int32_t> idx;
    double *buf = nullptr;                  // lamb | qty | betahat | Qt | moments
    fokl_host_job *job = nullptr;
    // G2 on the device (fokl_dspectral_*): buf is the job's page-locked result area until the ticket is released
    fokl_dspectral *dev = nullptr;
    int64_t ticket = 0;
    bool dev_waited = false;
    // G2 from the eigenpairs of the model with one more column (fokl_pool_submit_spectral_update): `parent` is kept until
    // this job has run; depth = how many such steps separate this model from a fresh decomposition
    Spectrum *parent = nullptr;
    int depth = 0;
    // the Gram the job reads, when the search owns it (kill tests decided at once: their G2 jobs outlive the call -- and
    // the caller's array -- that submitted them); jobs submitted through fokl_search_spectral* read the caller's
    std::shared_ptr<const std::vector<double>> gram_keep;
    // G2 not requested yet (launch_deferred): an accepted kill test's model in a sub-stage of wide models, where a
    // decomposition costs tens of milliseconds -- it is requested when something turns out to need it (a guessed decision
    // to confirm, the sub-stage's surviving model), and never for a model that is replaced before anything looked at it
    bool deferred = false;
    const double *deferred_gram = nullptr;
    int deferred_ld = 0;
    int32_t updated = -1;
    int status = FOKL_OK;
    int refs = 1;
    // (X'X)^-1 of the model, formed from the eigenpairs the first time a least-squares predictor starts from this model
    // (PathModel::init: p^3 operations, 0.1-0.25 ms at 100-144 columns) and kept: a sub-stage model is started from two or
    // three times -- the likely first tests before and after its evaluation, the kill tests themselves (search thread only)
    mutable std::vector<double> inverse;
    double *lamb() const { return buf; }
    double *qty() const { return buf + p1; }
    double *betahat() const { return buf + 2 * p1; }
    double *Qt() const { return buf + 3 * p1; }
    double *moments() const { return buf + 3 * p1 + (size_t)p1 * p1; }
};

struct Check {
    double value;
    bool decision;
};

// The statistics of a sub-stage model's new terms (FR:1656-1658), formed by the chain thread the moment its chain has run
// (round 6: the driver used to form them when it got there, 0.1-0.2 ms per sub-stage on the thread everything waits for).
// The new terms are the model's last `count` active columns.
struct TermStats {
    int count = 0, p1 = 0, draws = 0, half0 = 0, half1 = 0;
    const double *w = nullptr;              // the chain's draws in the eigenbasis [draws][p1]
    const double *Qt = nullptr;             // the model's eigenvectors (row e = eigenvector e); kept alive by the chain's owner
    std::vector<double> mean_abs, rel_std;
};

struct Outcome : Counted<Outcome, &Census::outcomes> {
    Spectrum *spec = nullptr;
    Tape *tape = nullptr;
    double ev = 0, siglik = 0, dtd = NAN;
    // host chain
    fokl_host_job *chain = nullptr;
    double *w = nullptr;
    size_t w_classes = 0;
    bool w_pinned = false;
    int32_t *flag = nullptr;                // bstar < 0 seen (written by the chain thread)
    bool chain_waited = false;
    int chain_status = FOKL_OK;
    // device chain
    bool on_device = false;
    int64_t ticket = 0;
    const double *stats_area = nullptr;     // page-locked: [flag, sig, tau, rows, mean w [p1], ticket, seconds]
    bool device_released = false;
    double intercept_scale = NAN;
    bool scale_inline = false;              // ... formed in line by the search thread (a borderline guess), not by verify()
    std::vector<Check> checks;
    bool release_wanted = false, released = false;
    // A kill test decided from the downdated least-squares model (fokl_search_set_decide, mode 1): the outcome exists from
    // the moment of the decision, its G2 may still be running and its chain has not been started -- `lazy` until
    // settle_pending has started the chain (or found that nobody will ever look at it: `cancelled`)
    bool lazy = false, cancelled = false;
    double s1 = NAN, s2 = NAN;              // residual moments the BIC was formed from
    double ev_replaced = NAN;               // the BIC this (accepted) test's had to be below (direct decisions)
    double ls_intercept = NAN;              // least-squares intercept known at decision time (lazy outcomes)
    double guess_margin = NAN;              // relative distance kept when guessing from it (NaN: the search's own)
    int64_t trace_index = -1;               // its record in fokl_search::trace
    int new_terms = 0;                      // a sub-stage model: its last new_terms columns are the new terms
    TermStats *tstats = nullptr;            // ... and their statistics, if a host chain thread forms them
    int refs = 1;                           // Python handle + the search's own lists
};

struct Forecast {
    std::vector<int32_t> key;               // device slots of the surviving model columns
    Spectrum *spec;
    int columns;
    double dtd;
    int likely_tests = -1;                  // kill tests its least-squares fit makes likely (computed once G2 is there)
};

enum Stat {
    S_GIBBS_CALLS, S_KILL_TESTS, S_TERMS_LOGICAL, S_T_EIGH, S_T_CHAIN, S_CHAINS_MATERIALISED, S_BIC_FROM_GRAM,
    S_TAPES_REWOUND, S_TAPES_WASTED, S_CHAINS_AHEAD, S_CHAINS_AHEAD_UNUSED, S_CHAINS_SKIPPED, S_SPECTRAL_SUBMITTED,
    S_DEVICE_CHAINS, S_CHAINS_FETCHED, S_GUESSED, S_GUESS_WAITS, S_GUESSES_VERIFIED, S_DCHAIN_KERNEL_S, S_DCHAIN_TIMED,
    S_T_RESID, S_T_KILL_LOOP, S_TAPES_MATERIALISED, S_ROWS_CHAINS, S_PATH_REPREDICTED, S_SPECTRAL_DEVICE, S_SPECTRAL_UPDATED,
    S_DIRECT_TESTS, S_DIRECT_MAX_REL, S_CHAINS_CANCELLED, S_T_SETTLE, S_GUESS_MAX_DEV, S_DIRECT_IN_BAND, S_STATS_BY_CHAIN_THREAD,
    S_COUNT
};

}  // namespace

struct fokl_search {
    fokl_host_pool *pool = nullptr;
    fokl_dchain *dchain = nullptr;
    fokl_dspectral *dspec = nullptr;        // G2 on the device for models of up to dspec_max columns (fokl_search_bind_spectral)
    int dspec_max = 0;
    bool dspec_staged = false, dspec_hold = false;
    // which jobs go there: those whose result is not wanted before the device can have it -- slack (microseconds until a
    // kill test will ask for the model) >= dspec_slack * the kernels' duration at that size; dspec_slack = 0: all of them.
    // The rest stay with LAPACK on the pool's threads (0.2-0.4 ms at 66 columns against the device's 0.6).
    double dspec_slack = 1.5;
    int dspec_lookahead = 32;               // how far ahead of the kill tests device jobs are requested
    double test_us = 70.0;                  // running mean of the time one kill test takes this search
    // kill tests' G2 from their parent model's eigenpairs (fokl_search_set_update): parents of update_from columns or more,
    // at most update_depth steps away from a fresh decomposition (each step waits for the one before it: the chain of
    // accepted tests along the predicted path is cut into pieces that the spectral threads work on side by side)
    int update_from = 0, update_depth = 3;
    // With derived models the look-ahead of the kill tests' G2 is this deep below kWideModel columns (0: prm.lookahead): a
    // chain of derivations advances one link per 0.05-0.2 ms, slower than the loop tests (40 us), and only the pieces that
    // the look-ahead window holds run side by side.  Not for the wide models: there a decomposition takes milliseconds and
    // a deeper window orders tapes further ahead than the stream keeps pre-states for (configs[3]: 24 deep 1.06 s against
    // 0.69, 48 deep fails).
    int lookahead_derived = 0;
    fokl_search_params prm{};
    double sigsqd0 = 0, tausqd0 = 0;
    int speculation = 0;
    int flip_guess = 0;
    bool pinned_tapes = false;
    // tapes on order, in stream order, no verdict yet
    std::deque<Tape *> spec;
    struct {
        Tape *tape = nullptr;
        Spectrum *spec = nullptr;
        fokl_host_job *job = nullptr;
        double *w = nullptr;
        size_t w_classes = 0;
        bool w_pinned = false;
        int32_t *flag = nullptr;
        TermStats *tstats = nullptr;
    } prechain;
    std::map<std::vector<int64_t>, double> ev_cache;
    std::vector<int64_t> active_ids;        // term id of every active column of the sub-stage (column 0: the intercept)
    std::deque<Outcome *> unverified;
    // Kill tests' BIC decisions (fokl_search_set_decide).  0: from G2 of the trial model (round 4: the loop waits for the
    // eigenpairs of every model it tests).  1: from the least-squares model of the sub-stage downdated column by column
    // (PathModel: SSR without column c = SSR + b_c^2 / [(X'X)^-1]_cc, exact in exact arithmetic) -- the decision takes
    // microseconds, G2 is requested for ACCEPTED models only and feeds nothing but their chains, which start when it
    // arrives (`pending`, in decision order); the BIC G2 brings along (Gram identity in 80-bit) is held against the
    // decision's, and a difference beyond direct_tolerance (relative) ends the search as a misprediction does.
    int decide = 0;
    double direct_tolerance = 1e-9;
    // Several ranks repeat this search side by side and must take every decision alike (fokl_search_set_deterministic):
    // nothing may then hinge on WHEN a chain's statistics arrive.  A second clause that can be guessed is guessed even if
    // the chain has run already (the guess is a function of the Gram alone), and a guess that its chain does not confirm
    // ends the search only where every rank finds out at the same point: in the blocking verification at the end.
    bool deterministic = false;
    bool misprediction_noted = false;
    // derivations per decomposition in sub-stages of wide models (kWideDepth; FOKL_EIGH_UPDATE_DEPTH_WIDE for experiments)
    int wide_depth = std::getenv("FOKL_EIGH_UPDATE_DEPTH_WIDE") ? std::max(1, std::atoi(std::getenv("FOKL_EIGH_UPDATE_DEPTH_WIDE"))) : 2;
    int defer_from = std::getenv("FOKL_G2_DEFER_FROM") ? std::atoi(std::getenv("FOKL_G2_DEFER_FROM")) : 192;   // (kWideModel)
    // accepted models waiting for G2 at most (the loop then waits for the oldest).  The device expands a tape from the
    // stream's pre-states, which the bulk threads keep for the last 1024 segments of 79 872 doubles (fokl_dchain_prestate_ring;
    // the device's own ring of regenerated segments is as long): a chain must be issued before the walker is that far past
    // its tape, so the bound follows the tape length -- 448 segments' worth of accepted models' tapes, 320 of tapes on order
    // (speculate), together well inside the 1024
    size_t pending_limit(int p1) const
    {
        const double per_tape = (double)prm.draws * (1.3 * p1 + 4.0) / 79872.0 + 1.0;    // segments (polar method: 1.27 doubles per normal)
        return (size_t)std::min(256.0, std::max(6.0, 448.0 / per_tape));
    }
    size_t order_limit(int p1) const
    {
        const double per_tape = (double)prm.draws * (1.3 * p1 + 4.0) / 79872.0 + 1.0;
        return (size_t)std::min(64.0, std::max(4.0, 320.0 / per_tape));
    }
    std::deque<Outcome *> pending;
    // FOKL_SEARCH_PROFILE=1: where the kill-test loop's own time goes (seconds per section, printed when the search ends)
    bool profile = std::getenv("FOKL_SEARCH_PROFILE") != nullptr;
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::deque<Outcome *> zombies;          // device chains nobody will look at: slots go back once they have run
    std::vector<Outcome *> device_outcomes;  // every device-chained outcome alive (all released when the search ends)
    std::vector<Forecast> forecasts;
    // what waits for threads before its memory can go
    std::vector<Tape *> tape_limbo;
    uint64_t tapes_requested = 0;
    struct WLimbo {                         // a host chain nobody looks at: what it reads and writes lives until it has run
        fokl_host_job *job;
        double *w;
        size_t classes;
        bool pinned;
        Tape *tape;
        int32_t *flag;
        Spectrum *spec;
        TermStats *tstats;
    };
    std::vector<WLimbo> chain_limbo;
    double stats[S_COUNT] = {};
    // 5 per evaluation: columns, built, ev, kill, mean intercept draw over the rows half0 .. (FR:1671; NaN until / unless the
    // search has looked at that chain's statistics)
    std::vector<double> trace;
    double last_siglik = 0;
    std::string error;
    bool mispredicted = false;
};

namespace {

int fail(fokl_search *s, int code, const std::string &msg)
{
    if (s) s->error = msg;
    fokl_set_global_error(msg);
    if (std::getenv("FOKL_SEARCH_PROFILE")) std::fprintf(stderr, "fokl_search: error %d: %s\n", code, msg.c_str());
    return code;
}

// (FOKL_SEARCH_PROFILE: where an error code that travels up the kill-test loop came from)
#define FOKL_RET(s_, x)                                                                                              \
    do {                                                                                                             \
        const int r__ = (x);                                                                                         \
        if (r__ != FOKL_OK && (s_)->profile) std::fprintf(stderr, "fokl_search: rc %d at line %d\n", r__, __LINE__);  \
        return r__;                                                                                                  \
    } while (0)

// ---- tapes ----------------------------------------------------------------------------------------------------

size_t pad8(size_t count) { return (count + 7) / 8 * 8; }

void carve(Tape *t)
{
    const size_t d = (size_t)t->draws, p1 = (size_t)t->p1, half = p1 / 2 + 1;
    const size_t nblocks = (d + FOKL_TAPE_BLOCK - 1) / FOKL_TAPE_BLOCK;
    const size_t ints0 = 16, ints1 = 16 + (nblocks + 15) / 16 * 16;
    size_t off[5];
    off[0] = 0;
    off[1] = off[0] + pad8(d * p1 + 16);
    off[2] = off[1] + pad8(d * half + 8);
    off[3] = off[2] + pad8(d);
    off[4] = off[3] + pad8(d);
    const size_t rows_at = off[4] + pad8((ints1 + d + 1) / 2) + 8;
    t->normals = t->mem;
    t->pair_r2 = t->mem + off[1];
    t->gam_sig = t->mem + off[2];
    t->gam_tau = t->mem + off[3];
    int32_t *ints = reinterpret_cast<int32_t *>(t->mem + off[4]);
    t->progress = ints;
    t->block_done = ints + ints0;
    t->lead = ints + ints1;
    t->rows = reinterpret_cast<fokl_tape_row *>(t->mem + rows_at);
    std::memset(ints, 0, ints1 * sizeof(int32_t));          // progress and block flags start at zero
}

size_t tape_doubles(int p1, int draws)
{
    const size_t d = (size_t)draws, half = (size_t)p1 / 2 + 1;
    const size_t nblocks = (d + FOKL_TAPE_BLOCK - 1) / FOKL_TAPE_BLOCK;
    const size_t ints1 = 16 + (nblocks + 15) / 16 * 16;
    return pad8(d * (size_t)p1 + 16) + pad8(d * half + 8) + 2 * pad8(d) + pad8((ints1 + d + 1) / 2) + 8 + 4 * d + 8;
}

void reap(fokl_search *s, bool block);

Tape *request_tape(fokl_search *s, int p1, bool tentative, bool model)
{
    auto *t = new Tape();
    t->seq = ++s->tapes_requested;
    t->p1 = p1;
    t->draws = s->prm.draws;
    t->model = model;
    t->tentative = tentative;
    const double astar = s->prm.a + 1 + s->prm.n / 2.0 + p1 / 2.0;      // FR:1508 (mmtx + 1 == p1)
    const double atau_star = s->prm.atau + (p1 - 1) / 2.0;            // FR:1510
    t->astar = astar;
    t->atau_star = atau_star;
    // a kill test's chain runs on the device when there is an engine: its tape stays raw (the device finishes it) -- or,
    // with the stream regenerated on the device, is never materialised on the host at all;
    // a model's tape is finished by host threads while it is walked (its statistics order the tests: latency matters)
    const bool on_device = !model && s->dchain && p1 <= s->prm.device_chain_columns;
    if (on_device && s->prm.device_rows && astar > 1.0 && atau_star > 1.0) {
        const size_t d = (size_t)t->draws;
        bool pinned = false;
        t->rows_mem = take_buffer(8 + 6 * d + 8, true, &t->rows_classes, &pinned);
        if (t->rows_mem && !pinned) {                       // the device reads rows in place: page-locked or not at all
            give_buffer(t->rows_mem, t->rows_classes, false);
            t->rows_mem = nullptr;
        }
        if (t->rows_mem) {
            t->rows_only = true;
            t->progress = reinterpret_cast<int32_t *>(t->rows_mem);
            std::memset(t->rows_mem, 0, 64);
            t->rows = reinterpret_cast<fokl_tape_row *>(t->rows_mem + 8);
            t->gam_sig = t->rows_mem + 8 + 4 * d;
            t->gam_tau = t->gam_sig + d;
            const int rc = fokl_pool_submit_noise(s->pool, p1, t->draws, astar, atau_star, t->rows, nullptr, nullptr, nullptr,
                                                  t->gam_sig, t->gam_tau, t->progress, tentative ? 1 : 0, nullptr,
                                                  FOKL_TAPE_BLOCK, 2, t->span, &t->noise);
            if (rc != FOKL_OK) {
                give_buffer(t->rows_mem, t->rows_classes, true);
                delete t;
                (void)fail(s, FOKL_ERR_STATE, "fokl_search: the pool refused a noise tape");
                return nullptr;
            }
            t->holds_stream = true;
            return t;
        }
    }
    t->mem = take_buffer(tape_doubles(p1, t->draws), s->pinned_tapes, &t->classes, &t->pinned);
    if (!t->mem) {
        delete t;
        fail(s, FOKL_ERR_STATE, "fokl_search: out of memory for a noise tape");
        return nullptr;
    }
    carve(t);
    t->finishing = !on_device && s->prm.finish_threads > 0;
    const int rc = fokl_pool_submit_noise(s->pool, p1, t->draws, astar, atau_star, t->rows, t->normals, t->pair_r2,
                                          t->lead, t->gam_sig, t->gam_tau, t->progress, tentative ? 1 : 0,
                                          t->block_done, FOKL_TAPE_BLOCK, t->finishing ? 1 : 0, nullptr, &t->noise);
    if (rc != FOKL_OK) {
        give_buffer(t->mem, t->classes, t->pinned);
        delete t;
        (void)fail(s, FOKL_ERR_STATE, "fokl_search: the pool refused a noise tape");
        return nullptr;
    }
    return t;
}

// A rows-only tape gets its arrays after all: a host chain has to read it (no device slot was free, a borderline guess is
// settled in line, the first row opens with the cached normal of the state handed over).  Waits for the walk.
int materialise(fokl_search *s, Tape *t)
{
    if (!t->rows_only) return FOKL_OK;
    for (int spins = 0;; ++spins) {
        const int32_t p = __atomic_load_n(t->progress, __ATOMIC_ACQUIRE);
        if (p < 0) return fail(s, FOKL_ERR_STATE, "fokl_search: the tape to materialise was sent back");
        if (p >= t->draws) break;
        if (spins < fokl_spin_budget(2000))
            _mm_pause();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(10));
    }
    t->mem = take_buffer(tape_doubles(t->p1, t->draws), false, &t->classes, &t->pinned);
    if (!t->mem) return fail(s, FOKL_ERR_STATE, "fokl_search: out of memory for a noise tape");
    const fokl_tape_row *rows = t->rows;
    const double *gs = t->gam_sig, *gt = t->gam_tau;
    int32_t *walked = t->progress;
    carve(t);
    std::memcpy(t->rows, rows, sizeof(fokl_tape_row) * (size_t)t->draws);
    std::memcpy(t->gam_sig, gs, sizeof(double) * (size_t)t->draws);       // variates the walker stored itself (shapes <= 1)
    std::memcpy(t->gam_tau, gt, sizeof(double) * (size_t)t->draws);
    (void)walked;
    const int rc = fokl_stream_expand(fokl_pool_stream(s->pool), t->p1, t->astar, t->atau_star, t->rows, 0, t->draws,
                                      t->normals, t->pair_r2, t->lead, t->gam_sig, t->gam_tau);
    if (rc != FOKL_OK) return rc;
    const int nblocks = (t->draws + FOKL_TAPE_BLOCK - 1) / FOKL_TAPE_BLOCK;
    for (int b = 0; b < nblocks; ++b) t->block_done[b] = 1;
    __atomic_store_n(t->progress, t->draws, __ATOMIC_RELEASE);
    t->rows_only = false;
    t->finishing = false;
    s->stats[S_TAPES_MATERIALISED] += 1;
    return FOKL_OK;
}

void unref(fokl_search *s, Tape *t)
{
    if (t && --t->refs == 0) s->tape_limbo.push_back(t);
}

void resolve(Tape *t, bool commit)
{
    if (t->tentative && !t->resolved) {
        t->resolved = true;
        (void)fokl_pool_resolve(t->noise, commit ? 1 : 0);
    }
}

// ---- spectra --------------------------------------------------------------------------------------------------

// Result buffers of host G2 jobs (lamb | qty | betahat | Qt | moments: 160 KB at 140 columns) go round in size classes,
// process-wide: fresh from malloc every one beyond 128 KB was its own mmap -- page-faulted in by the spectral thread that
// filled it, unmapped by the search thread that dropped it, ~400 times per fit.
constexpr int kSpectrumClass = 16;                          // columns per size class
std::mutex g_spectrum_m;
std::map<int, std::vector<double *>> g_spectrum_spares;     // class (columns rounded up) -> buffers
size_t g_spectrum_spare_bytes = 0;
constexpr size_t kSpectrumSpareMax = (size_t)256 << 20;

inline int spectrum_class(int p1) { return (p1 + kSpectrumClass - 1) / kSpectrumClass * kSpectrumClass; }
inline size_t spectrum_bytes(int cls) { return ((size_t)cls * (cls + 3) + 2) * sizeof(double); }

double *take_spectrum_buffer(int p1)
{
    const int cls = spectrum_class(p1);
    {
        std::lock_guard<std::mutex> lock(g_spectrum_m);
        auto it = g_spectrum_spares.find(cls);
        if (it != g_spectrum_spares.end() && !it->second.empty()) {
            double *buf = it->second.back();
            it->second.pop_back();
            g_spectrum_spare_bytes -= spectrum_bytes(cls);
            return buf;
        }
    }
    return static_cast<double *>(std::malloc(spectrum_bytes(cls)));
}

void give_spectrum_buffer(double *buf, int p1)
{
    if (!buf) return;
    const int cls = spectrum_class(p1);
    {
        std::lock_guard<std::mutex> lock(g_spectrum_m);
        if (g_spectrum_spare_bytes + spectrum_bytes(cls) <= kSpectrumSpareMax) {
            g_spectrum_spares[cls].push_back(buf);
            g_spectrum_spare_bytes += spectrum_bytes(cls);
            return;
        }
    }
    std::free(buf);
}

// how long the device takes for one decomposition, microseconds (profiles/eigh_device_r04.txt)
inline double device_spectral_us(int p1) { return p1 <= 72 ? 9.0 * p1 : (p1 <= 128 ? 14.0 * p1 : 28.0 * p1); }

// slack_us: microseconds until the result will be asked for, as far as the caller can tell (< 0: now)
inline bool spectrum_to_device(const fokl_search *s, int p1, double slack_us)
{
    if (!s->dspec || p1 > s->dspec_max) return false;
    return s->dspec_slack <= 0.0 || slack_us >= s->dspec_slack * device_spectral_us(p1);
}

void unref(fokl_search *s, Spectrum *sp);

// From this many columns on a decomposition takes milliseconds, the kill tests wait for G2 and nothing else, and the pool has
// its largest number of threads: chains of derived models are kept short there so that more of them run side by side
// (configs[3], 586-column models, 11 threads: depth 1 625 ms per fit, 2 637, 3 664, 6 688, unlimited 1127; none 694).
constexpr int kWideModel = 192, kWideDepth = 2;

// parent / parent_pos: the model this one is with its column number parent_pos deleted (a kill test's model and the model
// it is tested against), when the caller has it -- G2 then follows from the parent's eigenpairs where that is allowed.
Spectrum *submit_spectrum(fokl_search *s, const double *gram, int ld, const int32_t *idx, int p1, double slack_us = -1.0,
                          Spectrum *parent = nullptr, int parent_pos = -1, bool defer = false)
{
    auto *sp = new Spectrum();
    sp->p1 = p1;
    sp->idx.assign(idx, idx + p1);
    if (defer) {
        sp->deferred = true;
        sp->deferred_gram = gram;
        sp->deferred_ld = ld;
        return sp;
    }
    if (spectrum_to_device(s, p1, slack_us)) {
        // staged only: the entry point that ends this burst of requests launches them as one grid (flush_spectra)
        if (fokl_dspectral_submit(s->dspec, gram, ld, sp->idx.data(), p1, ld - 1, 0, &sp->ticket, &sp->buf) != FOKL_OK) {
            delete sp;
            (void)fail(s, FOKL_ERR_STATE, "fokl_search: the device refused a spectral job");
            return nullptr;
        }
        sp->dev = s->dspec;
        s->dspec_staged = true;
        s->stats[S_SPECTRAL_SUBMITTED] += 1;
        s->stats[S_SPECTRAL_DEVICE] += 1;
        return sp;
    }
    sp->buf = take_spectrum_buffer(p1);
    int rc = FOKL_ERR_STATE;
    const bool from_parent = sp->buf && parent && !parent->deferred && s->update_from > 0 && parent_pos >= 0 && parent_pos <= p1 &&
                             parent->p1 == p1 + 1 && parent->p1 >= s->update_from && !parent->dev && parent->buf &&
                             parent->status == FOKL_OK &&
                             parent->depth < (parent->p1 >= kWideModel ? std::min(s->update_depth, s->wide_depth) : s->update_depth);
    if (from_parent) {
        rc = fokl_pool_submit_spectral_update(s->pool, gram, ld, sp->idx.data(), p1, ld - 1, parent->lamb(), parent->Qt(),
                                              parent_pos, parent->job, sp->lamb(), sp->Qt(), sp->qty(), sp->betahat(),
                                              sp->moments(), &sp->updated, &sp->job);
        if (rc == FOKL_OK) {
            parent->refs += 1;                              // its arrays are read until this job has run
            sp->parent = parent;
            sp->depth = parent->depth + 1;
        }
    } else if (sp->buf) {
        rc = fokl_pool_submit_spectral(s->pool, gram, ld, sp->idx.data(), p1, ld - 1, sp->lamb(), sp->Qt(), sp->qty(),
                                       sp->betahat(), sp->moments(), &sp->job);
    }
    if (rc != FOKL_OK) {
        give_spectrum_buffer(sp->buf, p1);
        delete sp;
        (void)fail(s, FOKL_ERR_STATE, "fokl_search: the pool refused a spectral job");
        return nullptr;
    }
    s->stats[S_SPECTRAL_SUBMITTED] += 1;
    return sp;
}

// G2 of a deferred spectrum is wanted after all: a decomposition of its own (whatever it could have been derived from may
// never have been decomposed itself).
int launch_deferred(fokl_search *s, Spectrum *sp)
{
    if (!sp->deferred) return FOKL_OK;
    sp->buf = take_spectrum_buffer(sp->p1);
    int rc = FOKL_ERR_STATE;
    if (sp->buf)
        rc = fokl_pool_submit_spectral(s->pool, sp->deferred_gram, sp->deferred_ld, sp->idx.data(), sp->p1, sp->deferred_ld - 1,
                                       sp->lamb(), sp->Qt(), sp->qty(), sp->betahat(), sp->moments(), &sp->job);
    if (rc != FOKL_OK) {
        give_spectrum_buffer(sp->buf, sp->p1);
        sp->buf = nullptr;
        sp->status = fail(s, FOKL_ERR_STATE, "fokl_search: the pool refused a spectral job");
        return sp->status;
    }
    sp->deferred = false;
    s->stats[S_SPECTRAL_SUBMITTED] += 1;
    return FOKL_OK;
}

// The job of `sp` has run: its parent's arrays are no longer read; a model decomposed afresh after all is depth 0.
void job_has_run(fokl_search *s, Spectrum *sp)
{
    sp->job = nullptr;
    if (Spectrum *parent = sp->parent) {
        sp->parent = nullptr;
        if (sp->updated == 1)
            s->stats[S_SPECTRAL_UPDATED] += 1;
        else
            sp->depth = 0;
        unref(s, parent);
    }
}

// Launch the device jobs staged since the last call (one grid: they run side by side).
void flush_spectra(fokl_search *s)
{
    if (s->dspec_staged) {
        s->dspec_staged = false;
        (void)fokl_dspectral_flush(s->dspec);               // a failed launch is reported by the jobs' waits
    }
}

bool spectrum_done(Spectrum *sp)
{
    if (sp->deferred) return false;
    if (sp->dev) return sp->dev_waited || fokl_dspectral_poll(sp->dev, sp->ticket) != 0;   // (an error counts: wait reports it)
    return !sp->job || fokl_pool_poll(sp->job) != 0;
}

int wait_spectrum(fokl_search *s, Spectrum *sp)
{
    if (sp->deferred) {
        const int rc = launch_deferred(s, sp);
        if (rc != FOKL_OK) return rc;
    }
    if (sp->dev) {
        if (!sp->dev_waited) {
            const double t0 = now_s();
            sp->status = fokl_dspectral_wait(sp->dev, sp->ticket);
            sp->dev_waited = true;
            s->stats[S_T_EIGH] += now_s() - t0;
        }
        return sp->status;
    }
    if (sp->job) {
        const double t0 = now_s();
        sp->status = fokl_pool_wait(sp->job);
        job_has_run(s, sp);
        s->stats[S_T_EIGH] += now_s() - t0;
    }
    return sp->status;
}

void unref(fokl_search *s, Spectrum *sp)
{
    if (!sp || --sp->refs > 0) return;
    if (sp->dev) {
        (void)fokl_dspectral_release(sp->dev, sp->ticket);  // waits for a job still in flight
        delete sp;
        return;
    }
    if (sp->job) {
        (void)fokl_pool_wait(sp->job);                      // its buffers are written until it has run
        job_has_run(s, sp);
    }
    give_spectrum_buffer(sp->buf, sp->p1);
    delete sp;
}

// ---- chains and outcomes ----------------------------------------------------------------------------------------

double *take_w(fokl_search *s, int p1, size_t *classes, bool *pinned)
{
    return take_buffer((size_t)s->prm.draws * (size_t)p1, false, classes, pinned);
}

void drop_prechain(fokl_search *s)
{
    auto &pc = s->prechain;
    if (!pc.tape) return;
    // a chain started ahead that nobody will look at: its buffer goes back when it has run, and the tape it reads is
    // not reused before that (the chain's reference on the tape passes to the limbo entry)
    s->chain_limbo.push_back({pc.job, pc.w, pc.w_classes, pc.w_pinned, pc.tape, pc.flag, pc.spec, pc.tstats});
    pc = {};
    s->stats[S_CHAINS_AHEAD_UNUSED] += 1;
}

void term_stats_now(void *arg);

// new_terms > 0: the chain thread forms the statistics of the model's last new_terms columns behind the chain (*tstats_out)
fokl_host_job *submit_host_chain(fokl_search *s, Spectrum *sp, Tape *t, double dtd, double *w, int32_t *flag,
                                 int new_terms = 0, TermStats **tstats_out = nullptr)
{
    fokl_host_job *job = nullptr;
    TermStats *ts = nullptr;
    if (new_terms > 0 && new_terms < sp->p1 && tstats_out) {
        ts = new TermStats();
        ts->count = new_terms;
        ts->p1 = sp->p1;
        ts->draws = t->draws;
        ts->half0 = s->prm.half0;
        ts->half1 = (int)std::ceil(t->draws / 2.0 + 1.0);                   // FR:1656
        ts->w = w;
        ts->Qt = sp->Qt();
        if (ts->half1 >= ts->draws) {
            delete ts;
            ts = nullptr;
        }
    }
    const int rc = fokl_pool_submit_chain(s->pool, sp->lamb(), sp->qty(), sp->p1, s->prm.b, s->prm.btau, dtd, s->sigsqd0,
                                          s->tausqd0, t->draws, t->normals, t->pair_r2, t->lead, t->gam_sig, t->gam_tau,
                                          t->progress, t->block_done, FOKL_TAPE_BLOCK, t->finishing ? 1 : 0, w, flag,
                                          ts ? term_stats_now : nullptr, ts, &job);
    if (rc != FOKL_OK) {
        delete ts;
        return nullptr;
    }
    if (tstats_out) *tstats_out = ts;
    return job;
}

// Start the chain of the evaluation expected next -- G2 `sp`, p1 columns -- if its G2 has run and its tape is the oldest on
// order (engine.ForwardSelection._chain_ahead).
void chain_ahead(fokl_search *s, Spectrum *sp, int p1, double dtd, int new_terms = 0)
{
    if (s->spec.empty() || s->spec.front()->p1 != p1 || !spectrum_done(sp)) return;
    Tape *t = s->spec.front();
    if (s->dchain && !t->model) return;                     // a kill test's chain: submitted to the device at commit
    if (wait_spectrum(s, sp) != FOKL_OK) return;
    auto &pc = s->prechain;
    if (pc.tape) {
        if (pc.tape == t && pc.spec == sp) return;
        drop_prechain(s);
    }
    size_t classes;
    bool pinned;
    double *w = take_w(s, p1, &classes, &pinned);
    if (!w) return;
    auto *flag = new int32_t(0);
    TermStats *ts = nullptr;
    fokl_host_job *job = submit_host_chain(s, sp, t, dtd, w, flag, new_terms, &ts);
    if (!job) {
        give_buffer(w, classes, pinned);
        delete flag;
        return;
    }
    sp->refs += 1;
    t->refs += 1;                                           // the chain reads the tape whatever becomes of the order
    pc.tstats = ts;
    pc.tape = t;
    pc.spec = sp;
    pc.job = job;
    pc.w = w;
    pc.w_classes = classes;
    pc.w_pinned = pinned;
    pc.flag = flag;
    s->stats[S_CHAINS_AHEAD] += 1;
}

// engine.ForwardSelection._drop_speculation: send back the tapes on order beyond the first `keep`
void drop_speculation(fokl_search *s, size_t keep)
{
    while (s->spec.size() > keep) {
        Tape *t = s->spec.back();                           // youngest first
        s->spec.pop_back();
        if (s->prechain.tape == t) drop_prechain(s);
        const bool begun = __atomic_load_n(t->progress, __ATOMIC_ACQUIRE) > 0;
        resolve(t, false);
        unref(s, t);
        s->stats[S_TAPES_REWOUND] += 1;
        if (begun) {
            s->stats[S_TAPES_WASTED] += 1;
            // (a wrong order used to stall the loop behind the walker; with kill tests decided at once -- mode 1 -- the walker
            // only ever walks a wrong tape in time it would have idled: the book stays as deep as it is allowed to be)
            if (s->decide != 1) s->speculation = std::max(1, s->speculation - 2);
        }
    }
}

// engine.ForwardSelection._tape_for: the tape of the model evaluation that happens now
Tape *tape_for(fokl_search *s, int p1, bool model)
{
    if (!s->spec.empty()) {
        Tape *t = s->spec.front();
        if (t->p1 == p1) {
            s->spec.pop_front();
            resolve(t, true);
            s->speculation = std::min(s->prm.speculation_max, s->speculation + 1);
            return t;
        }
        drop_speculation(s, 0);
    }
    return request_tape(s, p1, false, model);
}

// engine.ForwardSelection._speculate: orders that agree with `sizes` stay, the others are sent back, missing ones are
// placed until s->speculation tapes are on order
void speculate(fokl_search *s, const std::vector<std::pair<int, bool>> &sizes)
{
    if (!s->prm.tentative_tapes) return;
    size_t k = 0;
    while (k < s->spec.size() && k < sizes.size() && s->spec[k]->p1 == sizes[k].first) ++k;
    // (kill tests decided at once: a caller that names fewer models than are on order -- a model's own evaluation knows of
    // its first test only -- does not take back what a better informed one ordered across the sub-stage boundary; a tape
    // that turns out wrong is sent back when its turn comes and costs the walker nothing it had to do instead)
    if (s->decide == 1 && k == sizes.size() && k <= s->spec.size()) return;
    if (k < s->spec.size()) drop_speculation(s, k);
    // (tapes on order hold the stream from where they begin, like the accepted models' tapes that wait for G2: the book is
    // as deep as the search allows or as half of pending_limit's segments are long, whichever is less)
    const size_t by_length = sizes.empty() ? 4 : s->order_limit(sizes.front().first);
    const size_t upto = std::max(k, std::min((size_t)s->speculation, by_length));
    for (size_t i = k; i < sizes.size() && i < upto; ++i) {
        if (s->prm.test_rewinds) {                          // tests: a recorded tape that is then discarded
            Tape *bogus = request_tape(s, sizes[i].first + 1, true, true);
            if (bogus) {
                while (__atomic_load_n(bogus->progress, __ATOMIC_ACQUIRE) < bogus->draws) std::this_thread::yield();
                resolve(bogus, false);
                unref(s, bogus);
                s->stats[S_TAPES_REWOUND] += 1;
            }
        }
        Tape *t = request_tape(s, sizes[i].first, true, sizes[i].second);
        if (!t) return;
        s->spec.push_back(t);
    }
}

int ensure_started(fokl_search *s, Outcome *o);
int settle_pending(fokl_search *s, bool block, Outcome *upto);
void note_intercept(fokl_search *s, const Outcome *o, double mean);

bool chain_done(Outcome *o)
{
    if (o->lazy) return false;                              // its chain has not even been started (settle_pending)
    if (o->on_device) {
        if (o->device_released) return true;
        double seen;
        __atomic_load(o->stats_area + 4 + o->spec->p1, &seen, __ATOMIC_ACQUIRE);
        return seen == (double)o->ticket;
    }
    return o->chain_waited || !o->chain || fokl_pool_poll(o->chain) != 0;
}

int wait_host_chain(fokl_search *s, Outcome *o)
{
    if (!o->chain_waited) {
        const double t0 = now_s();
        o->chain_status = o->chain ? fokl_pool_wait(o->chain) : FOKL_OK;
        if (o->chain_status != FOKL_OK && s->profile)
            std::fprintf(stderr, "fokl_search: host chain failed with %d: %s\n", o->chain_status, fokl_last_error(nullptr));
        o->chain = nullptr;
        o->chain_waited = true;
        s->stats[S_T_CHAIN] += now_s() - t0;
        s->stats[S_CHAINS_MATERIALISED] += 1;
        if (o->chain_status == FOKL_OK && o->flag && *o->flag)
            o->chain_status = fail(s, FOKL_ERR_NUMERIC,
                                   "bstar < 0 inside the Gibbs chain (only possible with b <= 0): the noise tape "
                                   "cannot reproduce the reference's skipped draw (FR:1538-1539)");
    }
    return o->chain_status;
}

// mean over the rows from `first_row` on of the intercept draws (betas[:, 0] = w Q[0, :]')
int mean_intercept_draw(fokl_search *s, Outcome *o, int first_row, double *out)
{
    if (s->profile && o->on_device == false && o->chain == nullptr && !o->chain_waited && !o->lazy)
        std::fprintf(stderr, "fokl_search: mean_intercept_draw on an outcome without a chain (cancelled %d released %d)\n",
                     (int)o->cancelled, (int)o->released);
    if (o->lazy || o->cancelled) {
        const int rc = ensure_started(s, o);
        if (rc != FOKL_OK) return rc;
    }
    const int p1 = o->spec->p1;
    const double *q0 = o->spec->Qt();                       // Qt[j][0] at j * p1
    if (o->on_device) {
        std::vector<double> st((size_t)4 + p1);
        const double t0 = now_s();
        const int rc = fokl_dchain_wait(s->dchain, o->ticket, st.data());
        s->stats[S_T_CHAIN] += now_s() - t0;
        if (rc != FOKL_OK) return rc;
        s->stats[S_DCHAIN_KERNEL_S] += o->stats_area[5 + p1];
        s->stats[S_DCHAIN_TIMED] += 1;
        if (st[0] != 0.0)
            return fail(s, FOKL_ERR_NUMERIC, "bstar < 0 inside the Gibbs chain (only possible with b <= 0)");
        double acc = 0.0;
        for (int j = 0; j < p1; ++j) acc += st[(size_t)4 + j] * q0[(size_t)j * p1];
        *out = acc;
        return FOKL_OK;
    }
    const int rc = wait_host_chain(s, o);
    if (rc != FOKL_OK) return rc;
    const int draws = s->prm.draws;
    double total = 0.0;
    for (int k = first_row; k < draws; ++k) {
        const double *row = o->w + (size_t)k * p1;
        double acc = 0.0;
        for (int j = 0; j < p1; ++j) acc += row[j] * q0[(size_t)j * p1];
        total += acc;
    }
    *out = draws > first_row ? total / (double)(draws - first_row) : NAN;
    return FOKL_OK;
}

int intercept_scale(fokl_search *s, Outcome *o, double *out)
{
    if (std::isnan(o->intercept_scale)) {
        double m;
        const int rc = mean_intercept_draw(s, o, s->prm.half0, &m);
        if (rc != FOKL_OK) return rc;
        o->intercept_scale = std::fabs(m);
        note_intercept(s, o, m);
    }
    *out = o->intercept_scale;
    return FOKL_OK;
}

void release_device_job(fokl_search *s, Outcome *o)
{
    if (o->device_released) return;
    if (fokl_dchain_try_release(s->dchain, o->ticket)) {
        o->device_released = true;
        unref(s, o->tape);                                  // the device job was the tape's last reader
        o->tape = nullptr;
    } else {
        o->refs += 1;
        s->zombies.push_back(o);
    }
}

void destroy_outcome(fokl_search *s, Outcome *o)
{
    if (o->on_device) {
        if (!o->device_released) {
            (void)fokl_dchain_release(s->dchain, o->ticket);          // waits until the chain has run
            o->device_released = true;
        }
    } else if (o->w) {
        if (o->chain && !o->chain_waited) {                 // still running: buffers, tape and spectrum go when it has
            o->spec->refs += 1;
            s->chain_limbo.push_back({o->chain, o->w, o->w_classes, o->w_pinned, o->tape, o->flag, o->spec, o->tstats});
            o->tstats = nullptr;
            o->tape = nullptr;
        } else {
            give_buffer(o->w, o->w_classes, o->w_pinned);
            delete o->flag;
        }
        o->w = nullptr;
        o->flag = nullptr;
    } else {
        delete o->flag;
    }
    delete o->tstats;
    unref(s, o->spec);
    unref(s, o->tape);
    delete o;
}

void unref(fokl_search *s, Outcome *o)
{
    if (o && --o->refs == 0) destroy_outcome(s, o);
}

// GibbsOutcome.release: the draws of this model can no longer be looked at
void release_outcome(fokl_search *s, Outcome *o)
{
    if (o->released) return;
    if (o->lazy) {
        // no chain yet: with guessed decisions to confirm it will run and be released by verify(); without, settle_pending
        // finds it released and never starts it
        if (!o->checks.empty())
            o->release_wanted = true;
        else
            o->released = true;
        return;
    }
    if (o->on_device) {
        if (!o->checks.empty()) {                           // its statistics still have to confirm guessed decisions
            o->release_wanted = true;
            return;
        }
        o->released = true;
        release_device_job(s, o);
        return;
    }
    if (!o->checks.empty()) {
        // a kill test decided before its chain existed may have had its second clause guessed like a device chain's and
        // then got a host chain after all (every device slot alive): its statistics still have to confirm the guesses
        o->release_wanted = true;
        return;
    }
    o->released = true;
    if (o->w) {
        if (o->chain_waited || !o->chain) {
            give_buffer(o->w, o->w_classes, o->w_pinned);
            delete o->flag;
            unref(s, o->tape);
        } else {
            o->spec->refs += 1;
            // (the chain thread writes the new terms' statistics behind the chain: they stay alive as long as the job)
            s->chain_limbo.push_back({o->chain, o->w, o->w_classes, o->w_pinned, o->tape, o->flag, o->spec, o->tstats});
            o->tstats = nullptr;
            o->chain = nullptr;
            o->chain_waited = true;
            o->chain_status = FOKL_ERR_STATE;
        }
        o->flag = nullptr;
        o->tape = nullptr;
        o->w = nullptr;
    }
}

// A tape nobody refers to any more: waits until its noise job (walk + materialisation) has run -- which frees the job --,
// then its memory goes back.
void bury(fokl_search *s, Tape *t)
{
    (void)fokl_pool_wait(t->noise);
    if (t->holds_stream && t->span[0] != ~(uint64_t)0)
        (void)fokl_pool_release_hold(s->pool, t->span[0]);  // nobody expands its rows any more
    give_buffer(t->mem, t->classes, t->pinned);
    give_buffer(t->rows_mem, t->rows_classes, true);
    delete t;
}

// What has been waiting for pool threads: tapes whose noise job (walk + materialisation) has run and that no chain reads
// any more, chains nobody looks at.
void reap(fokl_search *s, bool block)
{
    for (size_t i = 0; i < s->chain_limbo.size();) {
        auto &c = s->chain_limbo[i];
        if (block || fokl_pool_poll(c.job)) {
            (void)fokl_pool_wait(c.job);
            give_buffer(c.w, c.classes, c.pinned);
            delete c.flag;
            delete c.tstats;
            Tape *t = c.tape;
            Spectrum *sp = c.spec;
            s->chain_limbo[i] = s->chain_limbo.back();
            s->chain_limbo.pop_back();
            unref(s, t);
            unref(s, sp);
        } else {
            ++i;
        }
    }
    for (size_t i = 0; i < s->tape_limbo.size();) {
        Tape *t = s->tape_limbo[i];
        if (block || fokl_pool_poll(t->noise)) {
            bury(s, t);
            s->tape_limbo[i] = s->tape_limbo.back();
            s->tape_limbo.pop_back();
        } else {
            ++i;
        }
    }
}

// ---- BIC ------------------------------------------------------------------------------------------------------

double ev_from_moments(fokl_search *s, double s1, double s2, int p1)
{
    const double n = (double)s->prm.n;
    const double siglik = s2 / n - (s1 / n) * (s1 / n);     // np.var(y - X betahat), FR:1551
    s->last_siglik = siglik;
    const double lik = siglik > 0 ? -(n / 2) * std::log(siglik) - (n - 1) / 2 : NAN;
    double ev = p1 * std::log(n) - 2 * lik;                 // FR:1553-1554
    if (s->prm.aic) ev = ev + (2 - std::log(n)) * p1;       // FR:1653-1654 / FR:1684-1685
    return ev;
}

// engine.ForwardSelection._same_model_same_ev: the first score of a model is the score of every later evaluation of it
double same_model_same_ev(fokl_search *s, const int32_t *idx, int p1, double ev)
{
    std::vector<int64_t> key((size_t)p1);
    for (int i = 0; i < p1; ++i) key[(size_t)i] = s->active_ids[(size_t)idx[i]];
    std::sort(key.begin(), key.end());
    return s->ev_cache.emplace(std::move(key), ev).first->second;
}

int64_t record(fokl_search *s, int p1, int n_prev, double ev, bool kill)
{
    s->stats[S_GIBBS_CALLS] += 1;
    s->stats[S_KILL_TESTS] += kill ? 1 : 0;
    s->stats[S_TERMS_LOGICAL] += p1 - n_prev;
    s->trace.insert(s->trace.end(), {(double)p1, (double)(p1 - n_prev), ev, kill ? 1.0 : 0.0, NAN});
    return (int64_t)s->trace.size() / 5 - 1;
}

void note_intercept(fokl_search *s, const Outcome *o, double mean)
{
    if (o->trace_index >= 0 && (size_t)(5 * o->trace_index + 4) < s->trace.size()) s->trace[(size_t)(5 * o->trace_index + 4)] = mean;
}

// engine.ForwardSelection._commit: the chain of the evaluation (o->spec, which has run; o->tape) -- device engine for a
// kill test's candidate if there is one, else a host chain thread (the one started ahead if it is this one)
int start_chain(fokl_search *s, Outcome *o, double dtd, bool test)
{
    Spectrum *sp = o->spec;
    Tape *t = o->tape;
    const int p1 = sp->p1;
    if (test && t->rows_only && t->rows[0].lead_source == FOKL_SOURCE_GIVEN && (t->rows[0].start & FOKL_ROW_LEAD) &&
        __atomic_load_n(t->progress, __ATOMIC_ACQUIRE) > 0) {
        // (the very first tape of a stream handed over with a cached normal: that value exists on the host only)
        const int rc = materialise(s, t);
        if (rc != FOKL_OK) FOKL_RET(s, rc);
    }
    if (test && t->rows_only) {
        const int rc = fokl_dchain_submit_rows(s->dchain, p1, t->draws, sp->lamb(), sp->qty(), s->prm.b, s->prm.btau, dtd,
                                               s->sigsqd0, s->tausqd0, t->astar, t->atau_star, t->rows, t->gam_sig,
                                               t->gam_tau, t->progress, t->span, s->prm.half0, &o->ticket, &o->stats_area);
        if (rc == FOKL_OK) {
            o->on_device = true;
            o->refs += 1;
            s->device_outcomes.push_back(o);
            s->stats[S_DEVICE_CHAINS] += 1;
            s->stats[S_ROWS_CHAINS] += 1;
            return FOKL_OK;
        }
        if (rc != FOKL_ERR_STATE) FOKL_RET(s, rc);                // FOKL_ERR_STATE: every slot is alive -> host chain
        const int rcm = materialise(s, t);
        if (rcm != FOKL_OK) FOKL_RET(s, rcm);
    }
    if (test && s->dchain && !t->rows_mem && p1 <= s->prm.device_chain_columns && s->prechain.tape != t) {
        const int rc = fokl_dchain_submit(s->dchain, p1, t->draws, sp->lamb(), sp->qty(), s->prm.b, s->prm.btau, dtd,
                                          s->sigsqd0, s->tausqd0, t->normals, t->lead, t->gam_sig, t->gam_tau, t->progress,
                                          t->block_done, FOKL_TAPE_BLOCK, t->finishing ? 1 : 0, s->prm.half0, &o->ticket,
                                          &o->stats_area);
        if (rc == FOKL_OK) {
            o->on_device = true;
            o->refs += 1;
            s->device_outcomes.push_back(o);
            s->stats[S_DEVICE_CHAINS] += 1;
            return FOKL_OK;
        }
        if (rc != FOKL_ERR_STATE) FOKL_RET(s, rc);                // FOKL_ERR_STATE: every slot is alive -> the host chain
    }
    auto &pc = s->prechain;
    if (pc.tape) {
        if (pc.tape == t && pc.spec == sp) {                // the chain started ahead is the model's chain
            o->chain = pc.job;
            o->w = pc.w;
            o->w_classes = pc.w_classes;
            o->w_pinned = pc.w_pinned;
            o->flag = pc.flag;
            o->tstats = pc.tstats;
            unref(s, pc.spec);
            unref(s, pc.tape);                              // the chain's reference: the outcome holds the tape now
            pc = {};
            return FOKL_OK;
        }
        drop_prechain(s);
    }
    o->w = take_w(s, p1, &o->w_classes, &o->w_pinned);
    o->flag = new int32_t(0);
    if (!o->w) return fail(s, FOKL_ERR_STATE, "fokl_search: out of memory for a chain's draws");
    o->chain = submit_host_chain(s, sp, t, dtd, o->w, o->flag, o->new_terms, &o->tstats);
    if (!o->chain) {
        o->chain_waited = true;
        return fail(s, FOKL_ERR_STATE, "fokl_search: the pool refused a chain");
    }
    return FOKL_OK;
}

Outcome *commit(fokl_search *s, Spectrum *sp, Tape *t, double dtd, bool test, int new_terms = 0)
{
    auto *o = new Outcome();
    o->new_terms = new_terms;
    o->spec = sp;
    sp->refs += 1;
    o->tape = t;                                            // takes over the caller's reference
    if (start_chain(s, o, dtd, test) != FOKL_OK) {
        destroy_outcome(s, o);
        return nullptr;
    }
    return o;
}

// ---- kill tests decided from the downdated least-squares model: outcomes whose chains wait for G2 --------------------

// BIC of the moments, nothing else (ev_from_moments also notes siglik for the caller)
double ev_only(const fokl_search *s, double s1, double s2, int p1)
{
    const double n = (double)s->prm.n;
    const double siglik = s2 / n - (s1 / n) * (s1 / n);
    const double lik = siglik > 0 ? -(n / 2) * std::log(siglik) - (n - 1) / 2 : NAN;
    double ev = p1 * std::log(n) - 2 * lik;
    if (s->prm.aic) ev = ev + (2 - std::log(n)) * p1;
    return ev;
}

// The oldest pending outcomes whose G2 has run (block: all of them, waiting): the BIC of the eigenpairs is held against the
// one the decision was taken from; the chain starts -- unless the model has been replaced since and no guessed decision
// hangs on its intercept: nobody will ever look at those draws (FR:1686-1690 keeps the accepted model's only).
// upto: stop once this outcome has been dealt with (NULL: no such limit).
int settle_pending(fokl_search *s, bool block, Outcome *upto)
{
    bool started = false;
    // (a caller that waits wants the chains it has just started to run now, not when their batch has filled or aged)
    auto flush = [&] {
        if (block && started && s->dchain) (void)fokl_dchain_flush(s->dchain);
    };
    while (!s->pending.empty()) {
        Outcome *o = s->pending.front();
        if (o->spec->deferred && o->released && o->checks.empty() && !o->release_wanted) {
            // replaced before anything needed its eigenpairs: neither G2 nor a chain
            s->pending.pop_front();
            o->lazy = false;
            o->cancelled = true;
            s->stats[S_CHAINS_CANCELLED] += 1;
            if (s->prechain.tape == o->tape) drop_prechain(s);
            unref(s, o->tape);
            o->tape = nullptr;
            const bool last = o == upto;
            unref(s, o);
            if (last) break;
            continue;
        }
        if (!block && !spectrum_done(o->spec)) break;
        const double t0 = now_s();
        const int rc = wait_spectrum(s, o->spec);
        s->stats[S_T_SETTLE] += now_s() - t0;
        s->pending.pop_front();
        int out = rc;
        if (rc == FOKL_OK) {
            const double *m = o->spec->moments();
            const double other = ev_only(s, m[0], m[1], o->spec->p1);
            const double rel = std::fabs(other - o->ev) / std::fabs(o->ev);
            if (rel > s->stats[S_DIRECT_MAX_REL]) s->stats[S_DIRECT_MAX_REL] = rel;
            // ... and the decision itself: the model was accepted because its BIC lay below the one it replaced
            const bool contradicted = !std::isnan(o->ev_replaced) && !(other < o->ev_replaced);
            if ((!(rel <= s->direct_tolerance) || contradicted) && m[1] > 1e-6 * o->dtd) {
                if (s->deterministic) {
                    // ranks that repeat this search side by side get here at different moments (whose G2 has arrived when
                    // is timing): every rank reports it where all of them are, in the blocking verification (verify)
                    s->misprediction_noted = true;
                } else {
                    s->mispredicted = true;
                    out = fail(s, FOKL_ERR_STATE, "kill test decided from a downdated least-squares fit whose BIC the "
                                                  "eigenpairs of the model do not confirm");
                }
            }
        }
        if (out == FOKL_OK && o->released && o->checks.empty() && !o->release_wanted) {
            o->cancelled = true;                            // replaced before anybody needed its draws
            s->stats[S_CHAINS_CANCELLED] += 1;
        } else if (out == FOKL_OK) {
            out = start_chain(s, o, o->dtd, true);
            started = started || (out == FOKL_OK && o->on_device);
        }
        o->lazy = false;
        if (out != FOKL_OK || o->cancelled) {
            if (s->prechain.tape == o->tape) drop_prechain(s);
            unref(s, o->tape);                              // walked all the same: the stream advances as the reference's
            o->tape = nullptr;
            if (out != FOKL_OK) o->chain_status = out, o->chain_waited = true;
        }
        const bool last = o == upto;
        unref(s, o);                                        // the list's reference
        if (out != FOKL_OK) {
            flush();
            FOKL_RET(s, out);
        }
        if (last) break;
    }
    flush();
    return FOKL_OK;
}

// The chain of a lazy outcome is needed now (its statistics, its draws): everything up to it is settled, waiting for G2.
int ensure_started(fokl_search *s, Outcome *o)
{
    if (!o->lazy) return o->cancelled ? fail(s, FOKL_ERR_STATE, "fokl_search: the draws of a replaced model were never formed")
                                      : FOKL_OK;
    return settle_pending(s, true, o);
}

// engine.ForwardSelection._second_clause_now: `value < threshav * |mean intercept draw of o|` if it can be had without
// waiting for a chain.  -> 1 / 0, -1: the caller has to wait for the chain, < -1: error (-2 + FOKL_ERR_*)
int second_clause_now(fokl_search *s, Outcome *o, double value)
{
    // a lazy outcome's chain will run on the device if there is an engine for it (start_chain)
    const bool device_chain = o->on_device || (o->lazy && s->dchain && o->tape &&
                                               (o->tape->rows_only || o->spec->p1 <= s->prm.device_chain_columns));
    // (side by side with other ranks: only what every rank knows at this point of ITS search -- a scale formed in line
    // below; not one that verify() happened to have fetched already, nor a chain that happens to have run)
    const bool timing_free = !s->deterministic || !device_chain;
    if ((!std::isnan(o->intercept_scale) && (timing_free || o->scale_inline)) || (timing_free && chain_done(o))) {
        double scale;
        const int rc = intercept_scale(s, o, &scale);
        if (rc != FOKL_OK) return -2 + rc;
        return value < s->prm.threshav * scale ? 1 : 0;
    }
    if (!device_chain) return -1;
    const double threshold = s->prm.threshav * std::fabs(o->lazy ? o->ls_intercept : o->spec->betahat()[0]);
    const double margin = std::isnan(o->guess_margin) ? s->prm.guess_margin : std::max(s->prm.guess_margin, o->guess_margin);
    if (!(threshold > 0.0) || !std::isfinite(threshold) || std::fabs(value - threshold) <= margin * threshold) {
        // too close to call from the guess, and the device's answer is milliseconds away: the chain once more, in line on
        // this thread (same tape, same arithmetic up to the last bit of log()).  The tape is still there: the device job
        // is its reader until it has run.
        if (std::isnan(o->dtd) || !o->tape) return -1;
        if (o->lazy) {                                      // the eigenpairs the chain runs on may still be on their way
            const int rcw = wait_spectrum(s, o->spec);
            if (rcw != FOKL_OK) return -2 + rcw;
        }
        s->stats[S_GUESS_WAITS] += 1;
        const double t0 = now_s();
        Tape *t = o->tape;
        const int p1 = o->spec->p1;
        if (t->rows_only) {
            const int rcm = materialise(s, t);
            if (rcm != FOKL_OK) return -2 + rcm;
        }
        size_t classes;
        bool pinned;
        double *w = take_w(s, p1, &classes, &pinned);
        if (!w) return -2 + FOKL_ERR_STATE;
        int32_t negative = 0;
        int rc;
        if (t->finishing)
            rc = fokl_gibbs_chain_from_finished_tape(o->spec->lamb(), o->spec->qty(), p1, s->prm.b, s->prm.btau, o->dtd,
                                                     s->sigsqd0, s->tausqd0, t->draws, t->normals, t->gam_sig, t->gam_tau,
                                                     t->block_done, FOKL_TAPE_BLOCK, w, nullptr, nullptr, &negative);
        else
            rc = fokl_gibbs_chain_from_raw_blocks(o->spec->lamb(), o->spec->qty(), p1, s->prm.b, s->prm.btau, o->dtd,
                                                  s->sigsqd0, s->tausqd0, t->draws, t->normals, t->pair_r2, t->lead,
                                                  t->gam_sig, t->gam_tau, t->block_done, FOKL_TAPE_BLOCK, w, &negative);
        if (rc == FOKL_OK && negative) rc = fail(s, FOKL_ERR_NUMERIC, "bstar < 0 inside the Gibbs chain");
        if (rc == FOKL_OK) {
            const double *q0 = o->spec->Qt();
            double total = 0.0;
            for (int k = s->prm.half0; k < t->draws; ++k) {
                double acc = 0.0;
                for (int j = 0; j < p1; ++j) acc += w[(size_t)k * p1 + j] * q0[(size_t)j * p1];
                total += acc;
            }
            o->intercept_scale = std::fabs(total / (double)(t->draws - s->prm.half0));
            o->scale_inline = true;
            note_intercept(s, o, total / (double)(t->draws - s->prm.half0));
        }
        give_buffer(w, classes, pinned);
        s->stats[S_T_CHAIN] += now_s() - t0;
        if (rc != FOKL_OK) return -2 + rc;
        return value < s->prm.threshav * o->intercept_scale ? 1 : 0;
    }
    bool decision = value < threshold;
    s->stats[S_GUESSED] += 1;
    if (s->flip_guess && (int)s->stats[S_GUESSED] == s->flip_guess) decision = !decision;      // tests
    if (o->checks.empty()) {
        o->refs += 1;
        s->unverified.push_back(o);
        if (o->spec->deferred && launch_deferred(s, o->spec) != FOKL_OK) return -2 + FOKL_ERR_STATE;    // its chain will be needed
    }
    o->checks.push_back({value, decision});
    return decision ? 1 : 0;
}

// engine.ForwardSelection._verify: confirm the decisions taken from guessed intercept scales against the chains' own
// statistics -- those that have arrived, or (block) all of them.  FOKL_ERR_STATE + s->mispredicted if one does not hold.
int verify(fokl_search *s, bool block)
{
    const double t_in = s->profile && block ? now_s() : 0.0;
    struct Note {                                           // (FOKL_SEARCH_PROFILE: what the blocking verification waited for)
        fokl_search *s;
        double t_in, t_settled = 0.0;
        size_t pending, unverified;
        ~Note()
        {
            if (t_in > 0.0)
                std::fprintf(stderr, "fokl_search: blocking verify: %zu pending, %zu unverified; eigenpairs %.3f ms, chains %.3f ms\n",
                             pending, unverified, 1e3 * (t_settled - t_in), 1e3 * (now_s() - t_settled));
        }
    } note{s, t_in, t_in, s->pending.size(), s->unverified.size()};
    if (!s->pending.empty()) {
        const int rc = settle_pending(s, block, nullptr);
        if (rc != FOKL_OK) return rc;
    }
    if (t_in > 0.0) note.t_settled = now_s();
    while (!s->zombies.empty()) {
        Outcome *z = s->zombies.front();
        if (!z->device_released) {
            if (fokl_dchain_try_release(s->dchain, z->ticket)) {
                z->device_released = true;
            } else if (block) {
                (void)fokl_dchain_release(s->dchain, z->ticket);
                z->device_released = true;
            } else {
                break;
            }
            unref(s, z->tape);
            z->tape = nullptr;
        }
        s->zombies.pop_front();
        unref(s, z);
    }
    while (!s->unverified.empty() && (block || chain_done(s->unverified.front()))) {
        Outcome *o = s->unverified.front();
        s->unverified.pop_front();
        double scale;
        const int rc = intercept_scale(s, o, &scale);
        if (rc != FOKL_OK) {
            unref(s, o);
            return rc;
        }
        {
            // how far the chain's mean intercept lies from the least-squares intercept the guesses were taken from (relative):
            // what the guess margin has to cover
            const double guess = std::fabs(!std::isnan(o->ls_intercept) ? o->ls_intercept : o->spec->betahat()[0]);
            if (guess > 0.0 && !o->checks.empty())
                s->stats[S_GUESS_MAX_DEV] = std::max(s->stats[S_GUESS_MAX_DEV], std::fabs(scale / guess - 1.0));
        }
        for (const Check &c : o->checks) {
            if ((c.value < s->prm.threshav * scale) != c.decision) {
                if (s->deterministic && !block) {           // every rank will see it in its blocking verification
                    s->misprediction_noted = true;
                    continue;
                }
                s->mispredicted = true;
                unref(s, o);
                return fail(s, FOKL_ERR_STATE,
                            "kill test decided from a guessed intercept scale that its chain does not confirm");
            }
            s->stats[S_GUESSES_VERIFIED] += 1;
        }
        o->checks.clear();
        if (o->release_wanted) release_outcome(s, o);
        unref(s, o);
    }
    if (block && s->misprediction_noted) {
        s->mispredicted = true;
        return fail(s, FOKL_ERR_STATE, "kill test decided from a guessed intercept scale or a downdated least-squares fit "
                                       "that the model's chain / eigenpairs do not confirm");
    }
    return FOKL_OK;
}

// Where a sub-stage's kill tests will go, without an eigen-decomposition per step: removing column c from a least-squares
// model raises the sum of squared residuals by exactly b_c^2 / [(X'X)^-1]_cc and leaves (X'X)^-1 and b one rank-one
// downdate away.  The loop keeps this model next to the real one -- started from the sub-stage model's spectrum, advanced
// with every accepted test -- and asks it which of the remaining proposals will be tested and which accepted, so that the
// G2 jobs and the tapes of the whole path can be ordered far ahead (a spectral job is 0.4 ms: every wrong guess about the
// next model used to stall the loop for that long).  It only ever PREDICTS: every decision is taken from the real BIC.
#define FOKL_SEARCH_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

// inv [p x p] = sum over eigenpairs of q q' / lamb (Qt row e = eigenvector e)
FOKL_SEARCH_CLONES void inverse_from_spectrum(const double *__restrict__ Qt, const double *__restrict__ lamb, int p,
                                              double *__restrict__ inv, double *__restrict__ scaled)
{
    std::memset(inv, 0, sizeof(double) * (size_t)p * p);
    for (int e = 0; e < p; ++e) {
        const double *__restrict__ q = Qt + (size_t)e * p;
        const double w = 1.0 / lamb[e];
        for (int i = 0; i < p; ++i) scaled[i] = q[i] * w;
        for (int i = 0; i < p; ++i) {
            double *__restrict__ out = inv + (size_t)i * p;
            const double qi = scaled[i];
            for (int j = 0; j < p; ++j) out[j] += qi * q[j];
        }
    }
}

// row `oi` of the downdated inverse from row `i` of the old one: dst[j'] = src[j] - f column[j], column `pos` skipped
// (dst may be src: every element is read before the place it goes to is written)
FOKL_SEARCH_CLONES void downdate_row(const double *src, const double *__restrict__ column, double f, int p, int pos,
                                     double *dst)
{
    for (int j = 0; j < pos; ++j) dst[j] = src[j] - f * column[j];
    for (int j = pos + 1; j < p; ++j) dst[j - 1] = src[j] - f * column[j];
}

struct PathModel {
    int p = 0, ld = 0;
    std::vector<int32_t> cols;              // active column of every model position
    std::vector<double> inv;                // (X'X)^-1, [ld x ld], rows / columns of removed positions compacted away
    std::vector<double> beta;
    double ssr = 0, s1 = 0;

    void init(const Spectrum *sp)
    {
        p = ld = sp->p1;
        cols.assign(sp->idx.begin(), sp->idx.end());
        beta.assign(sp->betahat(), sp->betahat() + p);
        ssr = sp->moments()[1];
        s1 = sp->moments()[0];
        if (sp->inverse.empty()) {
            sp->inverse.resize((size_t)p * p);
            std::vector<double> scaled((size_t)p);
            inverse_from_spectrum(sp->Qt(), sp->lamb(), p, sp->inverse.data(), scaled.data());
        }
        inv = sp->inverse;
    }

    int position(int32_t col) const
    {
        for (int i = 0; i < p; ++i)
            if (cols[(size_t)i] == col) return i;
        return -1;
    }

    double ssr_without(int pos) const { return ssr + beta[(size_t)pos] * beta[(size_t)pos] / inv[(size_t)pos * ld + pos]; }

    // How far the mean of the chain's intercept draws (draws of them) can be expected from the least-squares intercept,
    // relative, times 32: the draws scatter with sigma^2 [(X'X + 1/tau^2)^-1]_00 <= siglik [(X'X)^-1]_00 around a mean that
    // the prior shrinks by parts in 1e6 at these sizes -- the distance a guessed second clause of FR:1670 keeps from its
    // threshold (on top of the search's floor).
    double guess_margin_for(double n, int rows_averaged) const
    {
        const double b0 = std::fabs(beta[0]);
        if (!(b0 > 0.0) || !(ssr > 0.0) || rows_averaged < 1) return INFINITY;
        return 32.0 * std::sqrt((ssr / n) * inv[0] / (double)rows_averaged) / b0;
    }

    void remove(int pos)
    {
        const double pivot = inv[(size_t)pos * ld + pos], bc = beta[(size_t)pos];
        ssr += bc * bc / pivot;
        std::vector<double> column((size_t)p);
        for (int i = 0; i < p; ++i) column[(size_t)i] = inv[(size_t)i * ld + pos];
        // downdate, compacting row / column `pos` away as we go (row oi <= i is written after row i was read)
        int oi = 0;
        for (int i = 0; i < p; ++i) {
            if (i == pos) continue;
            const double f = column[(size_t)i] / pivot;
            downdate_row(inv.data() + (size_t)i * ld, column.data(), f, p, pos, inv.data() + (size_t)oi * ld);
            beta[(size_t)oi] = beta[(size_t)i] - f * bc;
            cols[(size_t)oi] = cols[(size_t)i];
            ++oi;
        }
        p -= 1;
    }
};

std::vector<int32_t> columns_without(int A, const std::vector<int32_t> &removed /* sorted */)
{
    std::vector<int32_t> idx;
    idx.reserve((size_t)A);
    size_t r = 0;
    for (int c = 0; c < A; ++c) {
        if (r < removed.size() && removed[r] == c) {
            ++r;
            continue;
        }
        idx.push_back(c);
    }
    return idx;
}

std::vector<int32_t> with_column(const std::vector<int32_t> &set, int32_t c)
{
    std::vector<int32_t> out(set);
    out.insert(std::upper_bound(out.begin(), out.end(), c), c);
    return out;
}

// engine.ForwardSelection._likely_first_tests: the kill tests a sub-stage will probably run, guessed from the
// least-squares fit of its model (its last n_new columns are new).  -> active-column indices in testing order
std::vector<int32_t> likely_first_tests(fokl_search *s, const Spectrum *sp, int n_new, double siglik)
{
    const int A = sp->p1;
    std::vector<std::pair<double, int32_t>> order;
    std::vector<double> guess_std((size_t)n_new);
    const double *Qt = sp->Qt(), *lamb = sp->lamb(), *bh = sp->betahat();
    for (int j = 0; j < n_new; ++j) {
        const int col = A - n_new + j;
        double acc = 0.0;
        for (int e = 0; e < A; ++e) acc += Qt[(size_t)e * A + col] * Qt[(size_t)e * A + col] / lamb[e];
        guess_std[(size_t)j] = std::sqrt(std::max(siglik, 0.0) * acc);
        order.push_back({std::fabs(bh[col]), (int32_t)j});
    }
    std::stable_sort(order.begin(), order.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    const double floor = std::min(s->prm.threshstda, s->prm.threshstdb);
    std::vector<int32_t> out;
    for (const auto &e : order)
        if (guess_std[(size_t)e.second] > floor * e.first) out.push_back(A - n_new + e.second);
    return out;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------

extern "C" int fokl_search_create(fokl_host_pool *pool, fokl_dchain *dchain, const fokl_search_params *params,
                                  fokl_search **out)
{
    if (!pool || !params || !out || params->draws < 1 || params->n < 1 || params->speculation_max < 1)
        return fail(nullptr, FOKL_ERR_ARG, "fokl_search_create: null pointer or bad parameters");
    auto *s = new fokl_search();
    s->pool = pool;
    s->dchain = dchain;
    s->prm = *params;
    s->sigsqd0 = params->b / (1 + params->a);               // FR:1371
    s->tausqd0 = params->btau / (1 + params->atau);         // FR:1372
    s->speculation = params->speculation_max;
    s->flip_guess = params->flip_guess;
    s->pinned_tapes = dchain != nullptr;
    *out = s;
    return FOKL_OK;
}

// Everything still on order is sent back, everything still running is waited for, every buffer goes back to the store.
extern "C" void fokl_search_destroy(fokl_search *s)
{
    if (!s) return;
    double lap[5] = {0.0, 0.0, 0.0, 0.0, 0.0};             // FOKL_SEARCH_PROFILE: where the teardown's time goes
    double t_lap = s->profile ? now_s() : 0.0;
    auto mark = [&](int i) {
        if (!s->profile) return;
        const double t = now_s();
        lap[i] += t - t_lap;
        t_lap = t;
    };
    drop_speculation(s, 0);
    if (s->prechain.tape) drop_prechain(s);
    mark(0);
    while (!s->pending.empty()) {                           // chains that were never started: their tapes were walked, that is all
        Outcome *o = s->pending.front();
        s->pending.pop_front();
        o->lazy = false;
        o->cancelled = true;
        o->checks.clear();
        unref(s, o->tape);
        o->tape = nullptr;
        unref(s, o);
    }
    for (auto &f : s->forecasts) unref(s, f.spec);
    s->forecasts.clear();
    while (!s->unverified.empty()) {
        Outcome *o = s->unverified.front();
        s->unverified.pop_front();
        o->checks.clear();
        unref(s, o);
    }
    mark(1);
    (void)verify(s, true);                                  // zombies
    mark(2);
    for (Outcome *o : s->device_outcomes) {
        if (!o->device_released) {
            (void)fokl_dchain_release(s->dchain, o->ticket);
            o->device_released = true;
        }
        unref(s, o);
    }
    s->device_outcomes.clear();
    mark(3);
    reap(s, true);
    reap(s, true);                                          // tapes released by the chains of the first pass
    mark(4);
    if (s->profile)
        std::fprintf(stderr, "fokl_search teardown (ms): orders sent back %.3f, pending / forecasts / unverified let go %.3f, zombies %.3f, "
                             "device slots %.3f, waiting for jobs in flight %.3f\n", 1e3 * lap[0], 1e3 * lap[1], 1e3 * lap[2],
                     1e3 * lap[3], 1e3 * lap[4]);
    if (s->profile)
        std::fprintf(stderr, "fokl_search: alive after this search: %lld tapes, %lld spectra, %lld outcomes (handles the caller still holds)\n",
                     (long long)census().tapes.load(), (long long)census().spectra.load(), (long long)census().outcomes.load());
    if (s->profile)
        std::fprintf(stderr, "fokl_search profile (ms): verify %.3f clause %.3f tape %.3f bic %.3f accept %.3f (spectrum %.3f) "
                             "bounds %.3f tail %.3f\n", 1e3 * s->prof[0], 1e3 * s->prof[1], 1e3 * s->prof[2], 1e3 * s->prof[3],
                     1e3 * s->prof[4], 1e3 * s->prof[5], 1e3 * s->prof[6], 1e3 * s->prof[7]);
    delete s;
}

extern "C" const char *fokl_search_error(const fokl_search *s) { return s ? s->error.c_str() : ""; }

extern "C" int fokl_search_mispredicted(const fokl_search *s) { return s && s->mispredicted ? 1 : 0; }

extern "C" int fokl_search_set_substage(fokl_search *s, const int64_t *term_ids, int columns)
{
    if (!s || !term_ids || columns < 1) return fail(s, FOKL_ERR_ARG, "fokl_search_set_substage: bad arguments");
    s->active_ids.assign(term_ids, term_ids + columns);
    reap(s, false);
    return FOKL_OK;
}

extern "C" int fokl_search_speculate(fokl_search *s, const int32_t *sizes, const int32_t *is_model, int count)
{
    if (!s || (count > 0 && (!sizes || !is_model))) return fail(s, FOKL_ERR_ARG, "fokl_search_speculate: bad arguments");
    std::vector<std::pair<int, bool>> want;
    for (int i = 0; i < count; ++i) want.push_back({sizes[i], is_model[i] != 0});
    speculate(s, want);
    return FOKL_OK;
}

extern "C" int fokl_search_drop_speculation(fokl_search *s)
{
    if (!s) return fail(s, FOKL_ERR_ARG, "fokl_search_drop_speculation: null search");
    drop_speculation(s, 0);
    return FOKL_OK;
}

extern "C" int fokl_search_spectral(fokl_search *s, const double *gram, int ld, const int32_t *idx, int p1,
                                    fokl_spectrum **out)
{
    if (!s || !gram || !idx || !out || p1 < 1 || ld < 2) return fail(s, FOKL_ERR_ARG, "fokl_search_spectral: bad arguments");
    Spectrum *sp = submit_spectrum(s, gram, ld, idx, p1);
    if (!sp) return FOKL_ERR_STATE;
    *out = reinterpret_cast<fokl_spectrum *>(sp);
    if (!s->dspec_hold) flush_spectra(s);
    return FOKL_OK;
}

// The same for a model that is `parent`'s (a spectrum of this search, finished or not) without its column number parent_pos:
// G2 follows from the parent's eigenpairs where fokl_search_set_update allows it.
extern "C" int fokl_search_spectral_from(fokl_search *s, const double *gram, int ld, const int32_t *idx, int p1,
                                         fokl_spectrum *parent, int parent_pos, fokl_spectrum **out)
{
    if (!s || !gram || !idx || !out || p1 < 1 || ld < 2)
        return fail(s, FOKL_ERR_ARG, "fokl_search_spectral_from: bad arguments");
    Spectrum *sp = submit_spectrum(s, gram, ld, idx, p1, -1.0, reinterpret_cast<Spectrum *>(parent), parent_pos);
    if (!sp) return FOKL_ERR_STATE;
    *out = reinterpret_cast<fokl_spectrum *>(sp);
    if (!s->dspec_hold) flush_spectra(s);
    return FOKL_OK;
}

// G2 of models of up to max_columns columns goes to `engine` (NULL: back to the pool's LAPACK threads).
extern "C" int fokl_search_bind_spectral(fokl_search *s, fokl_dspectral *engine, int max_columns, double slack,
                                         int lookahead)
{
    if (!s) return fail(nullptr, FOKL_ERR_ARG, "fokl_search_bind_spectral: null search");
    s->dspec = engine;
    s->dspec_max = engine ? std::min(max_columns, fokl_dspectral_max_columns()) : 0;
    if (slack >= 0.0) s->dspec_slack = slack;
    if (lookahead >= 0) s->dspec_lookahead = lookahead;
    return FOKL_OK;
}

// G2 of the kill tests' models from their parent model's eigenpairs: parents of from_columns columns or more (0: never),
// at most `depth` such steps away from a fresh decomposition; `lookahead` (0: the search's own): how many tests ahead G2 is
// requested then, in sub-stages whose model has fewer than 192 columns.
extern "C" int fokl_search_set_update(fokl_search *s, int from_columns, int depth, int lookahead)
{
    if (!s) return fail(nullptr, FOKL_ERR_ARG, "fokl_search_set_update: null search");
    s->update_from = std::max(0, from_columns);
    s->update_depth = std::max(1, depth);
    s->lookahead_derived = std::max(0, lookahead);
    return FOKL_OK;
}

// How the kill tests' BICs are decided (see fokl_search::decide): 0 from G2 of every trial model, 1 from the sub-stage's
// least-squares model downdated column by column, confirmed by the accepted models' eigenpairs to `tolerance` (relative;
// <= 0: the default 1e-9).
extern "C" int fokl_search_set_decide(fokl_search *s, int mode, double tolerance)
{
    if (!s || mode < 0 || mode > 1) return fail(s, FOKL_ERR_ARG, "fokl_search_set_decide: bad arguments");
    s->decide = mode;
    if (tolerance > 0.0) s->direct_tolerance = tolerance;
    return FOKL_OK;
}

// Ranks that repeat one search side by side (see fokl_search::deterministic).
extern "C" int fokl_search_set_deterministic(fokl_search *s, int on)
{
    if (!s) return fail(nullptr, FOKL_ERR_ARG, "fokl_search_set_deterministic: null search");
    s->deterministic = on != 0;
    return FOKL_OK;
}

// Between hold(1) and hold(0) fokl_search_spectral only stages its jobs: hold(0) launches them as one grid.
extern "C" int fokl_search_hold_spectral(fokl_search *s, int hold)
{
    if (!s) return fail(nullptr, FOKL_ERR_ARG, "fokl_search_hold_spectral: null search");
    s->dspec_hold = hold != 0;
    if (!s->dspec_hold) flush_spectra(s);
    return FOKL_OK;
}

extern "C" int fokl_spectrum_done(fokl_spectrum *h) { return h && spectrum_done(reinterpret_cast<Spectrum *>(h)) ? 1 : 0; }

// lamb [p1] | qty [p1] | betahat [p1] | Qt [p1, p1] (row j = eigenvector j) | moments [2]
extern "C" int fokl_spectrum_wait(fokl_search *s, fokl_spectrum *h, const double **buffer, int *p1)
{
    if (!s || !h) return fail(s, FOKL_ERR_ARG, "fokl_spectrum_wait: null pointer");
    Spectrum *sp = reinterpret_cast<Spectrum *>(h);
    const int rc = wait_spectrum(s, sp);
    if (buffer) *buffer = sp->buf;
    if (p1) *p1 = sp->p1;
    return rc;
}

// one more reference on a spectrum of this search (fokl_spectrum_release when done with it)
extern "C" int fokl_spectrum_retain(fokl_search *s, fokl_spectrum *h)
{
    if (!s || !h) return fail(s, FOKL_ERR_ARG, "fokl_spectrum_retain: null pointer");
    reinterpret_cast<Spectrum *>(h)->refs += 1;
    return FOKL_OK;
}

extern "C" void fokl_spectrum_release(fokl_search *s, fokl_spectrum *h)
{
    if (s && h) unref(s, reinterpret_cast<Spectrum *>(h));
}

// A sub-stage's MODEL in two halves around the driver's K3 launch: begin = its tape (committed from the order, or
// requested), the tapes on order after it (`then`), G2 waited for; commit = its chain.  The BIC comes from the driver
// (fokl_search_score: residual moments of the device pass).
extern "C" int fokl_search_model_begin(fokl_search *s, const double *gram, int ld, const int32_t *idx, int p1,
                                       fokl_spectrum *given, const int32_t *then_sizes, const int32_t *then_model,
                                       int then_count, fokl_spectrum **spectrum_out, fokl_tape **tape_out)
{
    if (!s || !gram || !idx || p1 < 1 || !spectrum_out || !tape_out)
        return fail(s, FOKL_ERR_ARG, "fokl_search_model_begin: bad arguments");
    Tape *t = tape_for(s, p1, true);                        // requested first: it is walked while G2 runs
    if (!t) return FOKL_ERR_STATE;
    std::vector<std::pair<int, bool>> want;
    for (int i = 0; i < then_count; ++i) want.push_back({then_sizes[i], then_model[i] != 0});
    speculate(s, want);
    Spectrum *sp = reinterpret_cast<Spectrum *>(given);
    if (sp)
        sp->refs += 1;
    else
        sp = submit_spectrum(s, gram, ld, idx, p1);
    if (!sp) {
        unref(s, t);
        return FOKL_ERR_STATE;
    }
    flush_spectra(s);
    const int rc = wait_spectrum(s, sp);
    if (rc != FOKL_OK) {
        unref(s, sp);
        unref(s, t);
        return rc;
    }
    *spectrum_out = reinterpret_cast<fokl_spectrum *>(sp);
    *tape_out = reinterpret_cast<fokl_tape *>(t);
    return FOKL_OK;
}

extern "C" int fokl_search_model_commit(fokl_search *s, fokl_spectrum *spectrum, fokl_tape *tape, double dtd, int new_terms,
                                        fokl_outcome **out)
{
    if (!s || !spectrum || !tape || !out || new_terms < 0)
        return fail(s, FOKL_ERR_ARG, "fokl_search_model_commit: null pointer or negative count");
    Spectrum *sp = reinterpret_cast<Spectrum *>(spectrum);
    Outcome *o = commit(s, sp, reinterpret_cast<Tape *>(tape), dtd, false, new_terms);
    unref(s, sp);                                           // model_begin's reference: the outcome holds its own
    if (!o) return FOKL_ERR_STATE;
    *out = reinterpret_cast<fokl_outcome *>(o);
    return FOKL_OK;
}

// BIC of an evaluation from the residual moments (sum r, sum r^2) the driver measured; identical models score
// identically (the first score of a model stands); the evaluation is recorded in the trace.
extern "C" int fokl_search_score(fokl_search *s, fokl_outcome *h, double s1, double s2, int n_prev, int kill, double *ev)
{
    if (!s || !h || !ev) return fail(s, FOKL_ERR_ARG, "fokl_search_score: null pointer");
    Outcome *o = reinterpret_cast<Outcome *>(h);
    const int p1 = o->spec->p1;
    o->ev = same_model_same_ev(s, o->spec->idx.data(), p1, ev_from_moments(s, s1, s2, p1));
    o->siglik = s->last_siglik;
    o->s1 = s1;
    o->s2 = s2;
    o->trace_index = record(s, p1, n_prev, o->ev, kill != 0);
    *ev = o->ev;
    return FOKL_OK;
}

extern "C" int fokl_outcome_info(fokl_search *s, fokl_outcome *h, fokl_outcome_view *view)
{
    if (!s || !h || !view) return fail(s, FOKL_ERR_ARG, "fokl_outcome_info: null pointer");
    Outcome *o = reinterpret_cast<Outcome *>(h);
    view->p1 = o->spec->p1;
    view->on_device = o->on_device ? 1 : 0;
    view->ev = o->ev;
    view->siglik = o->siglik;
    view->intercept_scale = o->intercept_scale;
    view->spectrum = o->spec->buf;
    view->idx = o->spec->idx.data();
    return FOKL_OK;
}

// the outcome's G2 as a handle of its own (one more reference: fokl_spectrum_release when done with it)
extern "C" int fokl_outcome_spectrum(fokl_search *s, fokl_outcome *h, fokl_spectrum **out)
{
    if (!s || !h || !out) return fail(s, FOKL_ERR_ARG, "fokl_outcome_spectrum: null pointer");
    Outcome *o = reinterpret_cast<Outcome *>(h);
    o->spec->refs += 1;
    *out = reinterpret_cast<fokl_spectrum *>(o->spec);
    return FOKL_OK;
}

extern "C" int fokl_outcome_chain_ready(fokl_outcome *h) { return h && chain_done(reinterpret_cast<Outcome *>(h)) ? 1 : 0; }

// The draws in the eigenbasis, w [draws, p1] (betas = w Q'): waits for the chain; a device chain's draws are copied to
// the host (only models that are returned get here).  The memory belongs to the outcome.
extern "C" int fokl_outcome_draws(fokl_search *s, fokl_outcome *h, const double **w)
{
    if (!s || !h || !w) return fail(s, FOKL_ERR_ARG, "fokl_outcome_draws: null pointer");
    Outcome *o = reinterpret_cast<Outcome *>(h);
    if (o->released) return fail(s, FOKL_ERR_STATE, "fokl_outcome_draws: the outcome's draws were released");
    if (o->lazy || o->cancelled) {
        const int rc = ensure_started(s, o);
        if (rc != FOKL_OK) return rc;
    }
    if (o->on_device) {
        if (!o->w) {
            o->w = take_w(s, o->spec->p1, &o->w_classes, &o->w_pinned);
            if (!o->w) return fail(s, FOKL_ERR_STATE, "fokl_outcome_draws: out of memory");
            const double t0 = now_s();
            std::vector<double> st((size_t)4 + o->spec->p1);
            int rc = fokl_dchain_wait(s->dchain, o->ticket, st.data());
            if (rc == FOKL_OK && st[0] != 0.0) rc = fail(s, FOKL_ERR_NUMERIC, "bstar < 0 inside the Gibbs chain");
            if (rc == FOKL_OK) rc = fokl_dchain_fetch_w(s->dchain, o->ticket, o->w);
            s->stats[S_T_CHAIN] += now_s() - t0;
            if (rc != FOKL_OK) return rc;
            s->stats[S_CHAINS_FETCHED] += 1;
            s->stats[S_CHAINS_MATERIALISED] += 1;
        }
        *w = o->w;
        return FOKL_OK;
    }
    const int rc = wait_host_chain(s, o);
    if (rc != FOKL_OK) return rc;
    *w = o->w;
    return FOKL_OK;
}

// The statistics that order and gate a sub-stage's kill tests (FR:1656-1658) from the model's draws, without materialising
// betas = w Q': for the active columns `cols`, mean_abs = |mean over rows half1 .. of beta|, rel_std = (population) standard
// deviation over rows half1 .. / |mean over rows half0 ..| -- the reference's inconsistent pair of row ranges, on purpose.
// beta[k][c] = sum_j w[k][j] Qt[j][cols[c]], summed over j ascending; means and squared deviations row after row, as numpy's
// reduction along axis 0 does.  Waits for the chain.
#define FOKL_STATS_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
namespace {
FOKL_STATS_CLONES void betas_of_rows(const double *__restrict__ w, const double *__restrict__ B, int p1, int count, int k0,
                                     int k1, double *__restrict__ out)
{
    for (int k = k0; k < k1; ++k) {
        const double *__restrict__ row = w + (size_t)k * p1;
        double *__restrict__ b = out + (size_t)(k - k0) * count;
        for (int c = 0; c < count; ++c) b[c] = 0.0;
        for (int j = 0; j < p1; ++j) {
            const double wj = row[j];
            const double *__restrict__ q = B + (size_t)j * count;
            for (int c = 0; c < count; ++c) b[c] += wj * q[c];
        }
    }
}
}  // namespace

namespace {
// mean_abs / rel_std of the active columns `cols` from the draws w [draws][p1] and the eigenvectors Qt
void term_stats(const double *w, const double *Qt, int p1, int draws, const int32_t *cols, int count, int half0, int half1,
                double *mean_abs, double *rel_std)
{
    std::vector<double> B((size_t)p1 * count), betas((size_t)(draws - half0) * count), sum0((size_t)count, 0.0),
        sum1((size_t)count, 0.0), dev((size_t)count, 0.0);
    for (int j = 0; j < p1; ++j)
        for (int c = 0; c < count; ++c) B[(size_t)j * count + c] = Qt[(size_t)j * p1 + cols[c]];
    betas_of_rows(w, B.data(), p1, count, half0, draws, betas.data());
    for (int k = half0; k < draws; ++k) {
        const double *b = betas.data() + (size_t)(k - half0) * count;
        for (int c = 0; c < count; ++c) sum0[(size_t)c] += b[c];
        if (k >= half1)
            for (int c = 0; c < count; ++c) sum1[(size_t)c] += b[c];
    }
    const double rows0 = (double)(draws - half0), rows1 = (double)(draws - half1);
    for (int c = 0; c < count; ++c) sum1[(size_t)c] /= rows1;             // the mean over rows half1 ..
    for (int k = half1; k < draws; ++k) {
        const double *b = betas.data() + (size_t)(k - half0) * count;
        for (int c = 0; c < count; ++c) {
            const double d = b[c] - sum1[(size_t)c];
            dev[(size_t)c] += d * d;
        }
    }
    for (int c = 0; c < count; ++c) {
        mean_abs[c] = std::fabs(sum1[(size_t)c]);
        rel_std[c] = std::sqrt(dev[(size_t)c] / rows1) / std::fabs(sum0[(size_t)c] / rows0);
    }
}

// on the chain thread, behind a chain that has run (fokl_pool_submit_chain's `then`)
void term_stats_now(void *arg)
{
    auto *ts = static_cast<TermStats *>(arg);
    std::vector<int32_t> cols((size_t)ts->count);
    for (int c = 0; c < ts->count; ++c) cols[(size_t)c] = ts->p1 - ts->count + c;
    ts->mean_abs.resize((size_t)ts->count);
    ts->rel_std.resize((size_t)ts->count);
    term_stats(ts->w, ts->Qt, ts->p1, ts->draws, cols.data(), ts->count, ts->half0, ts->half1, ts->mean_abs.data(),
               ts->rel_std.data());
}
}  // namespace

extern "C" int fokl_outcome_new_term_stats(fokl_search *s, fokl_outcome *h, const int32_t *cols, int count, int half0,
                                           int half1, double *mean_abs, double *rel_std)
{
    if (!s || !h || !cols || count < 1 || !mean_abs || !rel_std || half0 < 0 || half1 < half0)
        return fail(s, FOKL_ERR_ARG, "fokl_outcome_new_term_stats: bad arguments");
    Outcome *o = reinterpret_cast<Outcome *>(h);
    const double *w = nullptr;
    const int rc = fokl_outcome_draws(s, h, &w);            // (waits for the chain -- and for what its thread formed behind it)
    if (rc != FOKL_OK) return rc;
    const int p1 = o->spec->p1, draws = s->prm.draws;
    if (half1 >= draws) return fail(s, FOKL_ERR_ARG, "fokl_outcome_new_term_stats: no rows to average");
    for (int c = 0; c < count; ++c)
        if (cols[c] < 0 || cols[c] >= p1) return fail(s, FOKL_ERR_ARG, "fokl_outcome_new_term_stats: column out of range");
    if (const TermStats *ts = o->tstats) {
        bool same = !ts->mean_abs.empty() && ts->count == count && ts->half0 == half0 && ts->half1 == half1 && ts->w == w;
        for (int c = 0; same && c < count; ++c) same = cols[c] == p1 - count + c;
        if (same) {
            std::memcpy(mean_abs, ts->mean_abs.data(), sizeof(double) * (size_t)count);
            std::memcpy(rel_std, ts->rel_std.data(), sizeof(double) * (size_t)count);
            s->stats[S_STATS_BY_CHAIN_THREAD] += 1;
            return FOKL_OK;
        }
    }
    term_stats(w, o->spec->Qt(), p1, draws, cols, count, half0, half1, mean_abs, rel_std);
    return FOKL_OK;
}

extern "C" int fokl_outcome_intercept_scale(fokl_search *s, fokl_outcome *h, double *scale)
{
    if (!s || !h || !scale) return fail(s, FOKL_ERR_ARG, "fokl_outcome_intercept_scale: null pointer");
    return intercept_scale(s, reinterpret_cast<Outcome *>(h), scale);
}

// The search will not look at this model's draws again (idempotent); the handle stays valid until fokl_outcome_drop.
extern "C" void fokl_outcome_release(fokl_search *s, fokl_outcome *h)
{
    if (s && h) release_outcome(s, reinterpret_cast<Outcome *>(h));
}

extern "C" void fokl_outcome_drop(fokl_search *s, fokl_outcome *h)
{
    if (!s || !h) return;
    Outcome *o = reinterpret_cast<Outcome *>(h);
    // (a device chain whose statistics still have to confirm guessed decisions keeps its slot until they have: the queue
    // of unverified outcomes holds a reference of its own)
    release_outcome(s, o);
    unref(s, o);
}

extern "C" int fokl_search_verify(fokl_search *s, int block)
{
    if (!s) return fail(s, FOKL_ERR_ARG, "fokl_search_verify: null search");
    return verify(s, block != 0);
}

// G2 of the coming sub-stage's model for one predicted set of survivors (key = their device slots): lets the kill-test
// loop order that model's first tests' tapes and start its chain across the sub-stage boundary.
extern "C" int fokl_search_register_forecast(fokl_search *s, const int32_t *key, int key_count, fokl_spectrum *spectrum,
                                             double dtd)
{
    if (!s || (key_count > 0 && !key) || !spectrum) return fail(s, FOKL_ERR_ARG, "fokl_search_register_forecast: bad arguments");
    Spectrum *sp = reinterpret_cast<Spectrum *>(spectrum);
    sp->refs += 1;
    s->forecasts.push_back({std::vector<int32_t>(key, key + key_count), sp, sp->p1, dtd});
    return FOKL_OK;
}

extern "C" void fokl_search_clear_forecasts(fokl_search *s)
{
    if (!s) return;
    for (auto &f : s->forecasts) unref(s, f.spec);
    s->forecasts.clear();
}

// The kill tests a sub-stage will probably run, guessed from the least-squares fit of its model before its chain's
// statistics are there (engine.ForwardSelection._likely_first_tests), in testing order -- and, per test, whether it will
// probably be accepted (PathModel: the BIC without the column from a rank-one formula), so that the caller can submit G2
// along the path the tests will really take.  accepted_out may be NULL.
extern "C" int fokl_search_likely_first_tests(fokl_search *s, fokl_spectrum *spectrum, int n_new, double siglik,
                                              int32_t *columns_out, int32_t *accepted_out, int *count)
{
    if (!s || !spectrum || !columns_out || !count) return fail(s, FOKL_ERR_ARG, "fokl_search_likely_first_tests: null pointer");
    Spectrum *sp = reinterpret_cast<Spectrum *>(spectrum);
    const int rc = wait_spectrum(s, sp);
    if (rc != FOKL_OK) return rc;
    if (std::isnan(siglik)) siglik = sp->moments()[1] / s->prm.n - (sp->moments()[0] / s->prm.n) * (sp->moments()[0] / s->prm.n);
    const auto cols = likely_first_tests(s, sp, n_new, siglik);
    for (size_t i = 0; i < cols.size(); ++i) columns_out[i] = cols[i];
    *count = (int)cols.size();
    if (accepted_out) {
        PathModel m;
        m.init(sp);
        double ev_floor = ev_from_moments(s, m.s1, m.ssr, m.p);
        for (size_t i = 0; i < cols.size(); ++i) {
            const int at = m.position(cols[i]);
            accepted_out[i] = 0;
            if (at <= 0) continue;
            const double ev = ev_from_moments(s, m.s1, m.ssr_without(at), m.p - 1);
            if (ev < ev_floor) {
                accepted_out[i] = 1;
                ev_floor = ev;
                m.remove(at);
            }
        }
    }
    return FOKL_OK;
}

extern "C" int fokl_search_stats(const fokl_search *s, double *values, int count)
{
    if (!s || !values || count < S_COUNT) return fail(nullptr, FOKL_ERR_ARG, "fokl_search_stats: bad arguments");
    std::memcpy(values, s->stats, sizeof(double) * S_COUNT);
    return S_COUNT;
}

extern "C" int64_t fokl_search_trace(const fokl_search *s, double *records, int64_t capacity)
{
    if (!s) return 0;
    const int64_t have = (int64_t)s->trace.size() / 5;
    if (records) std::memcpy(records, s->trace.data(), sizeof(double) * 5 * (size_t)std::min(have, capacity));
    return have;
}

// ---------------------------------------------------------------------------------------------------------------
// the kill tests of one sub-stage (FR:1666-1690)
// ---------------------------------------------------------------------------------------------------------------

extern "C" int fokl_search_kill_tests(fokl_search *s, const fokl_kill_tests_args *a, fokl_kill_tests_result *res)
{
    if (!s || !a || !res || !a->gram || !a->columns || !a->mean_abs || !a->rel_std || !a->slots || !a->best ||
        !res->killed)
        return fail(s, FOKL_ERR_ARG, "fokl_search_kill_tests: null pointer");
    const double t_begin = now_s();
    const int A = a->active, ld = A + 1, vm = a->proposals;
    const double *gram = a->gram;
    const double dtd = gram[(size_t)A * ld + A];
    const double threshav = s->prm.threshav;
    Outcome *best = reinterpret_cast<Outcome *>(a->best);
    best->refs += 1;                                        // this loop's own reference
    std::vector<char> clause1((size_t)vm);
    std::vector<int> proposal;
    for (int j = 0; j < vm; ++j) {
        clause1[(size_t)j] = a->rel_std[j] > s->prm.threshstdb;
        if (clause1[(size_t)j] || a->rel_std[j] > s->prm.threshstda) proposal.push_back(j);   // the others cannot pass FR:1670
    }
    // guess at "mean_abs < threshav * |mean intercept draw|" for proposals further down the list: the posterior mean of
    // the intercept is close to its least-squares value, and it barely moves from one accepted model to the next
    double scale_guess = std::fabs(best->spec->betahat()[0]);
    std::vector<int32_t> killed;                            // sorted active-column indices
    double evmin = best->ev;
    bool last_accepted = true;                              // predictor: proposals go the way the last went
    std::map<std::vector<int32_t>, Spectrum *> ahead;       // trial set -> G2 submitted ahead
    for (int i = 0; i < a->ahead_count; ++i) {
        Spectrum *sp = reinterpret_cast<Spectrum *>(a->ahead_spectra[i]);
        std::vector<int32_t> key(a->ahead_keys + a->ahead_offsets[i], a->ahead_keys + a->ahead_offsets[i + 1]);
        std::sort(key.begin(), key.end());
        sp->refs += 1;
        if (!ahead.emplace(std::move(key), sp).second) unref(s, sp);
    }
    bool idle_pending = a->idle_work != nullptr;
    int rc = FOKL_OK;

    auto likely = [&](int j) { return clause1[(size_t)j] || a->mean_abs[j] < threshav * scale_guess; };
    auto survivors_key = [&](const std::vector<int32_t> &pred) {
        std::vector<int32_t> key;
        size_t r = 0;
        for (int c = 1; c < A; ++c) {
            while (r < pred.size() && pred[r] < c) ++r;
            if (r < pred.size() && pred[r] == c) continue;
            key.push_back(a->slots[c]);
        }
        return key;
    };
    auto find_forecast = [&](const std::vector<int32_t> &pred) -> Forecast * {
        const auto key = survivors_key(pred);
        for (auto &f : s->forecasts)
            if (f.key == key) return &f;
        return nullptr;
    };
    // the predicted rest of the loop: for proposal[first_step + k] whether its test will run and be accepted
    struct Step {
        bool run, accept;
    };
    PathModel committed;                                    // the model accepted so far, as the predictor sees it
    committed.init(best->spec);
    std::vector<Step> path;
    size_t path_from = 0;                                   // path[k] <-> proposal[path_from + k]
    // (as far as the loop orders things ahead: G2 jobs `lookahead` tests deep, tapes speculation_max deep -- a downdate
    // is O(columns^2), the whole rest of a sub-stage of hundreds of columns would cost more than it saves)
    // G2 look-ahead of this sub-stage (see lookahead_derived)
    const int lookahead = s->update_from > 0 && s->lookahead_derived > 0 && A < kWideModel
                              ? std::max(s->prm.lookahead, s->lookahead_derived) : s->prm.lookahead;
    const int horizon = std::max(lookahead, s->prm.speculation_max) + 8;
    bool path_complete = false;                             // the path reaches the end of the proposals
    auto predict = [&](size_t pos) {
        PathModel m = committed;
        double ev_floor = evmin;
        path.clear();
        path_from = pos;
        int running = 0;
        size_t q = pos;
        for (; q < proposal.size() && running < horizon; ++q) {
            const int j = proposal[q];
            const double scale = std::fabs(m.beta[0]);      // the intercept of the model accepted so far (predicted)
            Step st{clause1[(size_t)j] || a->mean_abs[j] < threshav * (q == pos ? scale_guess : scale), false};
            if (st.run) {
                ++running;
                const int at = m.position(a->columns[j]);
                if (at > 0) {
                    const double ev = ev_from_moments(s, m.s1, m.ssr_without(at), m.p - 1);
                    st.accept = ev < ev_floor;
                    if (st.accept) {
                        ev_floor = ev;
                        m.remove(at);
                    }
                }
            }
            path.push_back(st);
        }
        path_complete = q == proposal.size();
    };
    auto step_at = [&](size_t q) -> Step {
        if (q >= path_from && q - path_from < path.size()) return path[q - path_from];
        return Step{likely(proposal[q]), true};            // beyond the horizon: the round-3 guess
    };
    bool foreseen_any = false;
    std::vector<int32_t> foreseen_last;
    auto forecast = [&](size_t pos) {
        // the kill set at the end of the loop if the rest goes as predicted
        if (!a->foresee || !path_complete) return;
        int rest = 0;
        for (size_t q = pos; q < proposal.size() && rest <= s->prm.foresight; ++q)
            if (step_at(q).run) ++rest;
        if (rest > s->prm.foresight) return;                // (most calls: nothing to tell yet)
        std::vector<int32_t> pred(killed);
        for (size_t q = pos; q < proposal.size(); ++q) {
            const Step st = step_at(q);
            if (st.run && st.accept) pred.push_back(a->columns[proposal[q]]);
        }
        std::sort(pred.begin(), pred.end());
        // (the callback is Python: told once per predicted kill set, not once per test -- once the coming sub-stage has been
        // built, idle_work, before which it can do nothing with the news)
        if (rest <= s->prm.foresight && (idle_pending || !foreseen_any || pred != foreseen_last)) {
            if (!idle_pending) {
                foreseen_any = true;
                foreseen_last = pred;
            }
            a->foresee(a->user, pred.data(), (int)pred.size());
        }
    };
    const bool direct_mode = s->decide == 1;
    auto order_tapes = [&](size_t pos) {
        // the tapes of what the stream serves next if the search goes on as predicted (see engine.py order_tapes)
        std::vector<std::pair<int, bool>> sizes;
        std::vector<int32_t> pred(killed);
        bool through = true;
        for (size_t q = pos; q < proposal.size(); ++q) {
            if ((int)sizes.size() >= s->prm.speculation_max) {
                through = false;
                break;
            }
            const Step st = step_at(q);
            if (st.run) {
                sizes.push_back({A - (int)pred.size() - 1, false});
                if (st.accept) pred.push_back(a->columns[proposal[q]]);
            }
        }
        if (through && a->vm_next >= 0) {
            std::sort(pred.begin(), pred.end());
            // across the boundary: the coming model, its first test (every first test is one column smaller whichever
            // proposal it removes) -- or, if G2 of that model is there already, all the tests its least-squares fit
            // makes likely
            // (kill tests that are decided at once, mode 1: as if every new term were tested and accepted until G2 of the
            // coming model says better -- a tape of the wrong size is rewound, an idle walker is lost time)
            int tests = direct_mode ? a->vm_next : std::min(a->vm_next, 1);
            if (Forecast *f = find_forecast(pred)) {
                if (f->likely_tests < 0 && spectrum_done(f->spec) && wait_spectrum(s, f->spec) == FOKL_OK) {
                    const double n = (double)s->prm.n;
                    const double *m = f->spec->moments();
                    f->likely_tests = (int)likely_first_tests(s, f->spec, a->vm_next, m[1] / n - (m[0] / n) * (m[0] / n)).size();
                }
                if (f->likely_tests >= 0) tests = f->likely_tests;
            }
            const int coming = A - (int)pred.size() + a->vm_next;
            sizes.push_back({coming, true});
            for (int t = 1; t <= tests; ++t) sizes.push_back({coming - t, false});
        }
        speculate(s, sizes);
        // ... and the chain of the very next evaluation, if its G2 is there
        int nxt = -1;
        for (size_t q = pos; q < proposal.size(); ++q)
            if (step_at(q).run) {
                nxt = proposal[q];
                break;
            }
        if (nxt >= 0) {
            auto it = ahead.find(with_column(killed, a->columns[nxt]));
            if (it != ahead.end()) chain_ahead(s, it->second, A - (int)killed.size() - 1, dtd);
        } else if (a->vm_next >= 0) {
            if (Forecast *f = find_forecast(killed))
                chain_ahead(s, f->spec, A - (int)killed.size() + a->vm_next, f->dtd, a->vm_next);
        }
    };
    // G2 of the models on the predicted path, `lookahead` tests deep beyond the one at `pos` (which is submitted whatever
    // the prediction says about it: it is about to be needed)
    auto submit_ahead = [&](size_t pos) -> int {
        std::vector<int32_t> cur(killed);
        int deep = 0;
        // host threads: `lookahead` tests deep; the device, which costs no CPU but answers later: as far as dspec_lookahead,
        // and only jobs it can finish before their test comes up
        const int far = s->dspec ? std::max(lookahead, s->dspec_lookahead) : lookahead;
        for (size_t q = pos; q < proposal.size() && deep <= far; ++q) {
            const Step st = step_at(q);
            if (q > pos && !st.run) continue;
            ++deep;
            auto key = with_column(cur, a->columns[proposal[q]]);
            if (ahead.find(key) == ahead.end()) {
                const auto idx = columns_without(A, key);
                const double slack = (deep - 1) * s->test_us;
                if (deep <= lookahead + 1 || spectrum_to_device(s, (int)idx.size(), slack)) {
                    // the model this test is held against: the current one, or the trial model of the accepted test before
                    Spectrum *parent = nullptr;
                    int parent_pos = -1;
                    if (s->update_from > 0) {
                        if (cur == killed)
                            parent = best->spec;
                        else if (auto it = ahead.find(cur); it != ahead.end())
                            parent = it->second;
                        if (parent) {
                            const auto pidx = columns_without(A, cur);
                            parent_pos = (int)(std::lower_bound(pidx.begin(), pidx.end(), a->columns[proposal[q]]) - pidx.begin());
                        }
                    }
                    Spectrum *sp = submit_spectrum(s, gram, ld, idx.data(), (int)idx.size(), slack, parent, parent_pos);
                    if (!sp) return FOKL_ERR_STATE;
                    ahead.emplace(key, sp);
                }
            }
            if (q == pos ? (st.run ? st.accept : true) : st.accept) cur = std::move(key);
        }
        flush_spectra(s);
        return FOKL_OK;
    };

    // ---- mode 1: decisions from the downdated least-squares model (fokl_search::decide) -------------------------------
    // Only where the sub-stage model's BIC came with its residual moments (fokl_search_score) and its Gram is not
    // numerically singular (there the eigen-solver's own choices decide, see fokl_host_pool::singular).
    const bool direct = s->decide == 1 && std::isfinite(best->s2) && best->spec->p1 == A &&
                        best->spec->lamb()[0] > 1e-9 * best->spec->lamb()[A - 1];
    auto run_direct = [&]() -> int {
        PathModel &m = committed;
        best->guess_margin = m.guess_margin_for((double)s->prm.n, s->prm.draws - s->prm.half0);
        // Wide models (a decomposition takes tens of milliseconds, derivations are cut after two steps): G2 of an accepted
        // model is requested when something needs it, not when the model is accepted (Spectrum::deferred)
        const bool defer_g2 = A >= s->defer_from;
        // G2 of the accepted models is not waited for in here: the jobs read the search's own copy of the Gram
        const auto own_gram = std::make_shared<const std::vector<double>>(gram, gram + (size_t)ld * ld);
        std::vector<Spectrum *> callers;                    // spectra taken over from the caller (they read ITS array)
        auto tests = [&]() -> int {
        m.ssr = best->s2;                                   // the device's residual pass, not the Gram identity
        m.s1 = best->s1;
        const double ssr_base = best->s2;
        const double best_cond = best->spec->lamb()[A - 1] / best->spec->lamb()[0];
        int rc2_end = FOKL_OK;
        double tp = s->profile ? now_s() : 0.0;
        auto lap = [&](int k) {
            if (s->profile) {
                const double t = now_s();
                s->prof[k] += t - tp;
                tp = t;
            }
        };
        for (size_t pos = 0; pos < proposal.size(); ++pos) {
            const int i = proposal[pos];
            bool decided = clause1[(size_t)i];
            lap(6);
            // (every fourth turn: a turn takes microseconds, chains and eigenpairs arrive by the millisecond)
            int rc2 = (pos & 3) == 0 ? verify(s, false) : FOKL_OK;      // (also starts the chains whose G2 has arrived)
            lap(0);
            if (rc2 != FOKL_OK) FOKL_RET(s, rc2);
            if (!decided) {
                const int quick = second_clause_now(s, best, a->mean_abs[i]);
                if (quick < -1) FOKL_RET(s, quick + 2);
                if (quick == 0) continue;
                if (quick == 1) {
                    decided = true;
                    if (!std::isnan(best->intercept_scale)) scale_guess = best->intercept_scale;
                }
            }
            if (!decided) {
                // no guess to be had (host chains; a borderline case whose tape is gone): the chain of `best` decides
                if ((rc2 = intercept_scale(s, best, &scale_guess)) != FOKL_OK) FOKL_RET(s, rc2);
                if (!(a->mean_abs[i] < threshav * scale_guess)) continue;
            }
            lap(1);
            const int32_t col = a->columns[i];
            const int at = m.position(col);
            if (at <= 0) return fail(s, FOKL_ERR_STATE, "fokl_search_kill_tests: a proposal is not a column of the model");
            const auto trial = with_column(killed, col);
            const int p1 = A - (int)trial.size();
            Tape *tape = tape_for(s, p1, false);            // the stream moves on at once
            if (!tape) FOKL_RET(s, FOKL_ERR_STATE);
            if (idle_pending) {
                idle_pending = false;
                if ((rc2 = a->idle_work(a->user)) != FOKL_OK) {
                    unref(s, tape);
                    FOKL_RET(s, rc2);
                }
            }
            lap(2);
            const auto idx = columns_without(A, trial);
            auto take_spectrum = [&](bool want_now = false) -> Spectrum * {
                const double t_sp = s->profile ? now_s() : 0.0;
                Spectrum *sp = nullptr;
                auto it = ahead.find(trial);
                if (it != ahead.end()) {
                    sp = it->second;                        // (the map's reference becomes the caller's)
                    ahead.erase(it);
                    if (!sp->gram_keep) {
                        sp->refs += 1;
                        callers.push_back(sp);
                    }
                } else {
                    sp = submit_spectrum(s, own_gram->data(), ld, idx.data(), p1, -1.0,
                                         s->update_from > 0 && !best->spec->deferred ? best->spec : nullptr, at,
                                         defer_g2 && !want_now);
                    if (sp) sp->gram_keep = own_gram;
                    flush_spectra(s);
                }
                if (s->profile) s->prof[5] += now_s() - t_sp;
                return sp;
            };
            double s1 = m.s1, s2 = m.ssr_without(at);
            Spectrum *sp = nullptr;
            if (!(s2 > 1e-6 * dtd)) {
                // a candidate that (nearly) interpolates the data: the device's residual pass on the eigenpairs' betahat,
                // as in mode 0
                if (!(sp = take_spectrum(true))) {
                    unref(s, tape);
                    FOKL_RET(s, FOKL_ERR_STATE);
                }
                rc2 = wait_spectrum(s, sp);
                const double t0 = now_s();
                if (rc2 == FOKL_OK && !a->residual)
                    rc2 = fail(s, FOKL_ERR_STATE, "fokl_search_kill_tests: a residual pass is needed and no callback was given");
                if (rc2 == FOKL_OK) rc2 = a->residual(a->user, sp->idx.data(), p1, sp->betahat(), &s1, &s2);
                if (rc2 != FOKL_OK) {
                    unref(s, sp);
                    unref(s, tape);
                    FOKL_RET(s, rc2);
                }
                s->stats[S_T_RESID] += now_s() - t0;
            } else {
                s->stats[S_BIC_FROM_GRAM] += 1;
            }
            double ev_now = ev_from_moments(s, s1, s2, p1);
            if (!sp) {
                // Too close to call from the downdate: the eigenpairs' BIC is only ever held against an ACCEPTED
                // decision's, to direct_tolerance -- so a comparison that a difference of that size could turn, plus what the
                // downdates since the sub-stage model have lost to its conditioning (each increment of the SSR to about
                // eps * cond; d BIC = n dSSR / SSR), is taken from the trial model's G2, as mode 0 takes all of them.
                const double cond = best_cond;
                const double band = s->direct_tolerance * std::fabs(ev_now) +
                                    (double)s->prm.n * 16.0 * 2.220446049250313e-16 * cond * std::fabs(s2 - ssr_base) / s2;
                if (std::fabs(ev_now - evmin) <= band) {
                    if (!(sp = take_spectrum(true))) {
                        unref(s, tape);
                        FOKL_RET(s, FOKL_ERR_STATE);
                    }
                    if ((rc2 = wait_spectrum(s, sp)) != FOKL_OK) {
                        unref(s, sp);
                        unref(s, tape);
                        FOKL_RET(s, rc2);
                    }
                    const double *mo = sp->moments();
                    s1 = mo[0];
                    s2 = mo[1];
                    ev_now = ev_from_moments(s, s1, s2, p1);
                    s->stats[S_DIRECT_IN_BAND] += 1;
                }
            }
            const double ev = same_model_same_ev(s, idx.data(), p1, ev_now);
            const double siglik = s->last_siglik;
            s->stats[S_DIRECT_TESTS] += 1;
            lap(3);
            const bool ev_accepted = ev < evmin;
            if (ev_accepted) {
                if (!sp && !(sp = take_spectrum())) {
                    unref(s, tape);
                    FOKL_RET(s, FOKL_ERR_STATE);
                }
                auto *cand = new Outcome();
                cand->spec = sp;                            // takes over this block's reference
                cand->tape = tape;
                cand->lazy = true;
                cand->dtd = dtd;
                cand->ev = ev;
                cand->siglik = siglik;
                cand->s1 = s1;
                cand->s2 = s2;
                cand->ev_replaced = evmin;
                m.remove(at);
                m.ssr = s2;                                 // (the residual pass's, where that decided)
                cand->ls_intercept = m.beta[0];
                cand->guess_margin = m.guess_margin_for((double)s->prm.n, s->prm.draws - s->prm.half0);
                cand->refs += 1;
                s->pending.push_back(cand);
                killed = trial;
                evmin = ev;
                release_outcome(s, best);                   // the model it replaces: its draws are history
                const bool own = best != reinterpret_cast<Outcome *>(a->best);
                unref(s, best);
                if (own) unref(s, best);                    // (nobody else holds an intermediate model's creation reference)
                best = cand;
                best->refs += 1;                            // this loop's reference (the creation reference becomes the caller's)
            } else {
                if (s->prechain.tape == tape) drop_prechain(s);
                unref(s, tape);                             // walked all the same: the stream advances as the reference's
                if (sp) unref(s, sp);
                s->stats[S_CHAINS_SKIPPED] += 1;
            }
            const int64_t rec = record(s, p1, a->n_prev, ev, true);
            if (ev_accepted) best->trace_index = rec;
            lap(4);
            // bounded: accepted models waiting for G2 (their tapes hold the stream), tapes waiting for the walker
            const size_t bound = s->pending_limit(A);
            while (s->pending.size() > bound)
                if ((rc2 = settle_pending(s, true, s->pending.front())) != FOKL_OK) FOKL_RET(s, rc2);
            if ((pos & 7) == 7) reap(s, false);
            while (s->tape_limbo.size() > 2 * bound) {
                reap(s, false);
                if (s->tape_limbo.size() <= 2 * bound) break;
                // the walker works through its queue in order: the oldest tape is the one it reaches first (reap() swaps
                // entries about, so the oldest is looked for, not assumed in front)
                auto oldest = std::min_element(s->tape_limbo.begin(), s->tape_limbo.end(),
                                               [](const Tape *x, const Tape *y) { return x->seq < y->seq; });
                Tape *t_old = *oldest;
                *oldest = s->tape_limbo.back();
                s->tape_limbo.pop_back();
                bury(s, t_old);
            }
        }
        // the model that survives the sub-stage is the caller's: its eigenpairs will be looked at
        if (best->lazy && best->spec->deferred && (rc2_end = launch_deferred(s, best->spec)) != FOKL_OK) FOKL_RET(s, rc2_end);
        // the kill set is final: G2 of the coming sub-stage's model can start
        lap(6);
        if (a->foresee && !idle_pending) {
            std::vector<int32_t> pred(killed);
            a->foresee(a->user, pred.data(), (int)pred.size());
        }
        lap(7);
        return FOKL_OK;
        };
        const int rc_tests = tests();
        // jobs that read the caller's Gram have run before this call returns
        for (Spectrum *sp : callers) {
            (void)wait_spectrum(s, sp);
            unref(s, sp);
        }
        return rc_tests;
    };
    if (direct) {
        rc = run_direct();
    } else {
    predict(0);
    forecast(0);
    order_tapes(0);
    double t_test = now_s();
    for (size_t pos = 0; pos < proposal.size() && rc == FOKL_OK; ++pos) {
        const int i = proposal[pos];
        bool decided = clause1[(size_t)i];
        {
            const double t = now_s(), us = 1e6 * (t - t_test);
            t_test = t;
            if (pos > 0) s->test_us = std::min(400.0, std::max(20.0, 0.9 * s->test_us + 0.1 * us));
        }
        if ((rc = verify(s, false)) != FOKL_OK) break;
        if (!path_complete && pos + (size_t)(horizon / 2) >= path_from + path.size()) {
            predict(pos);                                   // the window of predicted steps moves on
            order_tapes(pos);
        }
        if (!decided) {
            // the second clause without a wait: from the chain of `best` if it has run, from its least-squares intercept
            // (confirmed later) if that is a device chain and the proposal is not a borderline case
            const int quick = second_clause_now(s, best, a->mean_abs[i]);
            if (quick < -1) {
                rc = quick + 2;
                break;
            }
            if (quick == 0) {
                if (step_at(pos).run) {
                    predict(pos + 1);
                    s->stats[S_PATH_REPREDICTED] += 1;
                    order_tapes(pos + 1);
                }
                continue;
            }
            if (quick == 1) {
                decided = true;
                if (!std::isnan(best->intercept_scale)) scale_guess = best->intercept_scale;
            }
        }
        if (!decided && (!std::isnan(best->intercept_scale) || !likely(i))) {
            // second clause without G2 of a model that will probably not be needed: from the known scale, or -- the test
            // looks unlikely -- after waiting for the chain of `best`
            if ((rc = intercept_scale(s, best, &scale_guess)) != FOKL_OK) break;
            if (!(a->mean_abs[i] < threshav * scale_guess)) {
                if (step_at(pos).run) {
                    predict(pos + 1);
                    s->stats[S_PATH_REPREDICTED] += 1;
                    order_tapes(pos + 1);
                }
                continue;
            }
            decided = true;
        }
        if ((rc = submit_ahead(pos)) != FOKL_OK) break;
        const auto trial = with_column(killed, a->columns[i]);
        const int p1 = A - (int)trial.size();
        // the test runs for sure: its tape is committed (or, not on order after a wrong guess, requested) now, so that
        // the stream moves on while this thread waits for G2
        Tape *tape = decided ? tape_for(s, p1, false) : nullptr;
        if (decided && !tape) {
            rc = FOKL_ERR_STATE;
            break;
        }
        if (idle_pending) {
            idle_pending = false;
            if ((rc = a->idle_work(a->user)) != FOKL_OK) {
                unref(s, tape);
                break;
            }
        }
        Spectrum *sp = ahead[trial];
        ahead.erase(trial);
        if ((rc = wait_spectrum(s, sp)) != FOKL_OK) {
            unref(s, sp);
            unref(s, tape);
            break;
        }
        if (!decided) {
            if ((rc = intercept_scale(s, best, &scale_guess)) != FOKL_OK) {      // waits for the chain of `best`
                unref(s, sp);
                break;
            }
            if (!(a->mean_abs[i] < threshav * scale_guess)) {
                unref(s, sp);
                if (step_at(pos).run) {
                    predict(pos + 1);
                    s->stats[S_PATH_REPREDICTED] += 1;
                    order_tapes(pos + 1);
                }
                continue;
            }
            tape = tape_for(s, p1, false);
            if (!tape) {
                unref(s, sp);
                rc = FOKL_ERR_STATE;
                break;
            }
        }
        // The BIC comes from the Gram and is known now, before anything is spent on the candidate's draws: a rejected
        // candidate only has to advance the random stream (its tape is walked, never chained -- nobody reads the draws
        // of a model that loses, FR:1686-1690).  A candidate that (nearly) interpolates the data gets the device's
        // residual pass after all: y'y - 2 b'Xty + b'XtX b cancels y'y / SSR digits.
        double s1 = sp->moments()[0], s2 = sp->moments()[1];
        if (!(s2 > 1e-6 * dtd)) {
            const double t0 = now_s();
            if (!a->residual || (rc = a->residual(a->user, sp->idx.data(), p1, sp->betahat(), &s1, &s2)) != FOKL_OK) {
                if (rc == FOKL_OK) rc = fail(s, FOKL_ERR_STATE, "fokl_search_kill_tests: a residual pass is needed and no callback was given");
                unref(s, sp);
                unref(s, tape);
                break;
            }
            s->stats[S_T_RESID] += now_s() - t0;
        } else {
            s->stats[S_BIC_FROM_GRAM] += 1;
        }
        const double ev = same_model_same_ev(s, sp->idx.data(), p1, ev_from_moments(s, s1, s2, p1));
        const double siglik = s->last_siglik;
        Outcome *cand = nullptr;
        if (ev < evmin) {
            cand = commit(s, sp, tape, dtd, true);          // takes over the tape's reference
            if (!cand) {
                unref(s, sp);
                rc = FOKL_ERR_STATE;
                break;
            }
        } else {
            if (s->prechain.tape == tape) drop_prechain(s);
            unref(s, tape);                                 // walked all the same: the stream advances as the reference's
            s->stats[S_CHAINS_SKIPPED] += 1;
        }
        const int64_t rec = record(s, p1, a->n_prev, ev, true);
        if (cand) cand->trace_index = rec;
        last_accepted = ev < evmin;
        const Step foreseen = step_at(pos);
        if (last_accepted) {
            const int at = committed.position(a->columns[i]);
            if (at > 0) committed.remove(at);
        }
        if (last_accepted) {
            killed = trial;
            evmin = ev;
            release_outcome(s, best);                       // the model it replaces: its draws are history
            // (an accepted test's outcome that is replaced in its turn: nobody holds its creation reference either)
            const bool own = best != reinterpret_cast<Outcome *>(a->best);
            unref(s, best);
            if (own) unref(s, best);
            best = cand;
            best->ev = ev;
            best->siglik = siglik;
            best->dtd = dtd;
            best->refs += 1;                                // this loop's reference (the creation reference is the caller's)
        }
        unref(s, sp);
        if (!foreseen.run || foreseen.accept != last_accepted) {
            // the path parts from what was predicted: resynchronise the predictor with the real model and look again
            if (last_accepted) committed.init(best->spec);
            predict(pos + 1);
            s->stats[S_PATH_REPREDICTED] += 1;
        }
        forecast(pos + 1);
        order_tapes(pos + 1);
        if ((pos & 7) == 7) reap(s, false);
    }
    }
    if (rc == FOKL_OK) {
        order_tapes(proposal.size());                       // the kill set is final
        if (idle_pending) {
            rc = a->idle_work(a->user);
            if (rc == FOKL_OK && direct && a->foresee) {
                std::vector<int32_t> pred(killed);
                a->foresee(a->user, pred.data(), (int)pred.size());
            } else if (rc == FOKL_OK) {
                forecast(proposal.size());
            }
        }
    }
    for (auto &kv : ahead) unref(s, kv.second);
    reap(s, false);
    // the caller's handle on `best`: the outcome it passed in if no test was accepted, else a new one (whose creation
    // reference becomes the caller's)
    res->best = reinterpret_cast<fokl_outcome *>(best);
    res->best_is_new = best != reinterpret_cast<Outcome *>(a->best) ? 1 : 0;
    best->refs -= 1;                                        // this loop's reference
    res->killed_count = (int)killed.size();
    for (size_t k = 0; k < killed.size(); ++k) res->killed[k] = killed[k];
    res->evmin = evmin;
    s->stats[S_T_KILL_LOOP] += now_s() - t_begin;
    return rc;
}
