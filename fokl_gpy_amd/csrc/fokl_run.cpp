// The sub-stage loop of FoKL.fit (FoKLRoutines.py:1602-1748) next to the kill-test loop (round 6).
//
// engine.ForwardSelection._run -- the Python statement of this loop, and what drives every search this file does not --
// spends ~9 ms of a 25 ms configs[2] fit in the interpreter: 15 000 function calls between ~60 native calls per sub-stage,
// on the one thread everything else waits for.  fokl_run_* is the same loop as native code for the common case: one process,
// the native search (csrc/fokl_search.cpp) on a host pool, look-ahead on.  It enumerates the sub-stages' candidate terms
// (FR:1350-1354, 1602-1648, 1722-1747), builds their columns and Gram blocks one sub-stage ahead (K1 + K2), evaluates the
// sub-stage's model (G2 through the search, K3 on the device), orders the likely tests' work ahead, waits for the model's
// statistics (FR:1656-1664), runs the kill tests (fokl_search_kill_tests, FR:1666-1690) with the coming model's G2 foreseen
// from inside, commits the survivors (FR:1691-1695) and applies the stop rule (FR:1701-1721).  Python keeps the class
// surface, the set-up of pool / search / engines, the final confirmation of guessed decisions and the returned arrays;
// FOKL_SUBSTAGE_LOOP=python runs engine._run instead (tests hold the two against each other and both against the goldens).
//
// The device is reached through a table of entry points with the signatures of include/fokl_hip.h (fokl_backend_ops): the
// library's own functions on a fokl_ctx, or -- CPU tests -- callbacks into the checker backend.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/fokl_hip_internal.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip / fokl_host_only.cpp

namespace {

inline double now_s()
{
    return 1e-9 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
                      std::chrono::steady_clock::now().time_since_epoch())
                      .count();
}

using Gram = std::shared_ptr<std::vector<double>>;          // [(A + 1)^2], y last; shared with the G2 jobs that read it

// engine.distinct_arrangements: all distinct orderings of the multiset in ascending lexicographic order
std::vector<int32_t> arrangements(const std::vector<int32_t> &indvec)
{
    std::vector<int32_t> cur(indvec);
    std::sort(cur.begin(), cur.end());
    std::vector<int32_t> rows(cur);
    while (std::next_permutation(cur.begin(), cur.end())) rows.insert(rows.end(), cur.begin(), cur.end());
    return rows;
}

// engine.ForwardSelection._patterns: (stage, indvec) of every sub-stage in the reference's order
struct Patterns {
    int m = 0, n_phis = 0, sett = 1, ind = 1;
    bool way3 = false, fresh = true, done = false;
    std::vector<int32_t> indvec;

    void deal()                                             // engine.deal_indvec (FR:1605-1613)
    {
        indvec.assign((size_t)m, 0);
        const int q = ind / sett, r = ind % sett;
        for (int j = 0; j < sett && j < m; ++j) indvec[(size_t)j] = q + (j < r ? 1 : 0);
    }

    bool advance()                                          // engine.advance_indvec (FR:1722-1740)
    {
        if (m == 1) return false;
        auto &v = indvec;
        if (way3) {
            if (v[1] > v[2]) {
                v[0] += 1;
                v[1] -= 1;
            } else if (v[2]) {
                v[1] += 1;
                v[2] -= 1;
                if (v[1] > v[0]) {
                    v[0] += 1;
                    v[1] -= 1;
                }
            } else {
                return false;
            }
            return true;
        }
        if (v[1]) {
            v[0] += 1;
            v[1] -= 1;
            return true;
        }
        return false;
    }

    // -> false when the search has run out of sub-stages (FR:1747)
    bool next(int *stage, std::vector<int32_t> *out)
    {
        if (done) return false;
        if (fresh) {
            deal();
            fresh = false;
        } else if (!advance()) {
            ind += 1;
            if (ind > n_phis) {
                done = true;
                return false;
            }
            deal();
        }
        *stage = ind;
        *out = indvec;
        return true;
    }
};

// engine.SlotPool: free list of device column slots
struct Slots {
    const fokl_backend_ops *ops = nullptr;
    int capacity = 0;
    std::vector<int32_t> free_list;

    int grow(int cap)
    {
        if (cap <= capacity) return FOKL_OK;
        const int rc = ops->reserve_slots(ops->ctx, cap);
        if (rc != FOKL_OK) return rc;
        const int lo = std::max(capacity, FOKL_SLOT_FIRST_FREE);
        for (int s = cap - 1; s >= lo; --s) free_list.push_back(s);
        capacity = cap;
        return FOKL_OK;
    }

    int take(int count, std::vector<int32_t> *out)
    {
        if ((int)free_list.size() < count) {
            const int rc = grow(capacity + std::max(std::max(count - (int)free_list.size(), capacity / 2), 32));
            if (rc != FOKL_OK) return rc;
        }
        out->clear();
        for (int i = 0; i < count; ++i) {
            out->push_back(free_list.back());
            free_list.pop_back();
        }
        return FOKL_OK;
    }

    void give(std::vector<int32_t> slots)
    {
        std::sort(slots.begin(), slots.end(), [](int32_t x, int32_t y) { return x > y; });
        free_list.insert(free_list.end(), slots.begin(), slots.end());
    }
};

// a sub-stage built ahead: its terms, their slots, their Gram rows against [active columns of the sub-stage before | new | y]
struct Ahead {
    bool valid = false;
    std::vector<int32_t> vecs;                              // [T][m]
    int T = 0;
    std::vector<int32_t> slots;
    int over = 0;                                           // active columns of the sub-stage it was launched under
    bool pending = false;                                   // the Gram block is still on the device
    bool built = false;                                     // K1 went out, K2 not yet (valid stays false until it has)
    std::vector<double> block;                              // [T][over + T + 1]
};

struct SpectrumRef {                                        // one reference on a G2 job + the Gram it reads
    fokl_spectrum *h = nullptr;
    Gram gram;
};

// engine.ForwardSelection._extend_gram
Gram extend_gram(const std::vector<double> &gram, int A, const std::vector<int> &keep, const std::vector<double> &block, int vm,
                 int block_ld, const std::vector<int> &block_kept, int over)
{
    const int n_prev = (int)keep.size(), A2 = n_prev + vm, L = A2 + 1, old = A + 1;
    auto out = std::make_shared<std::vector<double>>((size_t)L * L);
    double *o = out->data();
    for (int i = 0; i < n_prev; ++i) {
        const double *src = gram.data() + (size_t)keep[(size_t)i] * old;
        for (int j = 0; j < n_prev; ++j) o[(size_t)i * L + j] = src[keep[(size_t)j]];
        o[(size_t)i * L + A2] = o[(size_t)A2 * L + i] = src[A];
    }
    o[(size_t)A2 * L + A2] = gram[(size_t)A * old + A];
    for (int r = 0; r < vm; ++r) {
        const double *b = block.data() + (size_t)r * block_ld;
        double *row = o + (size_t)(n_prev + r) * L;
        for (int j = 0; j < n_prev; ++j) {
            const double v = b[block_kept[(size_t)j]];
            row[j] = v;
            o[(size_t)j * L + n_prev + r] = v;
        }
        for (int c = 0; c < vm; ++c) row[n_prev + c] = b[over + c];
        row[A2] = o[(size_t)A2 * L + n_prev + r] = b[over + vm];
    }
    return out;
}

}  // namespace

struct fokl_run {
    fokl_backend_ops ops{};
    fokl_run_params prm{};
    Slots pool;
    Patterns patterns;
    // the head start (fokl_run_create): the seed Gram and the first sub-stage under way before the host threads exist
    std::vector<double> base;                               // Gram of [ones, y]
    Ahead ahead;
    bool have_next = false;                                 // (stage, indvec) of the sub-stage `ahead` belongs to / comes next
    int next_stage = 0;
    std::vector<int32_t> next_indvec;
    // results
    std::vector<int32_t> mtx;                               // [terms][m]: the returned model's interaction matrix
    int mtx_rows = 0;
    std::vector<double> evs;
    fokl_outcome *best_model = nullptr, *last_model = nullptr;     // handles the caller takes over (may be the same)
    std::vector<double> stat_mean_abs, stat_rel_std;
    std::vector<int32_t> stat_sizes;
    double stats[FOKL_RUN_STATS] = {};
    std::string error;
};

namespace {

int run_fail(fokl_run *r, int code, const std::string &msg)
{
    if (r) r->error = msg;
    fokl_set_global_error(msg);
    return code;
}

enum RunStat {
    R_TERMS_PHYSICAL, R_SUBSTAGES, R_FORECASTS_USED, R_FORECASTS_EARLY, R_RESID_MATRIX_FREE, R_T_RESID, R_PHASE_PREPARE,
    R_PHASE_MODEL, R_PHASE_STATISTICS, R_PHASE_TESTS, R_PHASE_WRAP_UP, R_COUNT
};
static_assert(R_COUNT <= FOKL_RUN_STATS, "fokl_run_stats: grow FOKL_RUN_STATS");

// engine.HipBackend.resid_terms_supported / resid_terms_pay_from: can a model made of these terms take the matrix-free
// residual pass, and from how many columns on does it pay?  -> -1: not supported
int matrix_free_pay_from(const fokl_backend_ops &ops, const std::vector<int32_t> &terms, int rows, int m)
{
    static const int layouts[8][2] = {{8, 1}, {16, 1}, {8, 2}, {4, 4}, {2, 8}, {8, 4}, {16, 2}, {4, 8}};
    if (!ops.bic_resid_terms_launch || rows < 1) return -1;
    int max_order = 0;
    for (int t = 0; t < rows; ++t) {
        int nz = 0;
        for (int j = 0; j < m; ++j) {
            const int v = terms[(size_t)t * m + j];
            nz += v != 0;
            max_order = std::max(max_order, v);
        }
        if (nz > 2) return -1;
    }
    if (max_order == 0) return -1;
    if (ops.kernel_id != 0 && max_order > 8) return -1;
    int inputs = 0, deepest = 0;
    for (int j = 0; j < m; ++j) {
        std::vector<int32_t> seen;
        for (int t = 0; t < rows; ++t) {
            const int v = terms[(size_t)t * m + j];
            if (v != 0 && std::find(seen.begin(), seen.end(), v) == seen.end()) seen.push_back(v);
        }
        inputs += !seen.empty();
        deepest = std::max(deepest, (int)seen.size());
    }
    int best = -1;
    for (auto &l : layouts)
        if (inputs <= l[0] && deepest <= l[1] && (best < 0 || l[0] * l[1] < best)) best = l[0] * l[1];
    if (best < 0) return -1;
    return ops.kernel_id != 0 ? 0 : (int)(3.5 * best);
}

// FOKL_BUILD_AHEAD_AT=model: K1 + K2 of the coming sub-stage are launched behind the model's evaluation (round 6's first form);
// default `start`: K1 (the coming columns) at the sub-stage's start, in front of the wait for the model's eigenpairs -- the device
// builds them while the host waits --, K2 (their Gram rows) right behind the model's residual pass in the device's queue: the
// coming model's G2 can then be requested a quarter of a millisecond earlier and is waited for that much less at the next
// sub-stage's start.  (Both in front of the residual pass: the pass, which the model's BIC waits for, queued behind them -- 1.8 ms
// per fit in t_resid for 1.3 gained in t_eigh.)
static const bool g_build_at_start = !(std::getenv("FOKL_BUILD_AHEAD_AT") && std::strcmp(std::getenv("FOKL_BUILD_AHEAD_AT"), "model") == 0);

// the loop's state
struct Loop {
    fokl_run *r;
    fokl_search *s;
    const fokl_backend_ops &ops;
    const fokl_run_params &prm;
    int m;
    // the model so far
    std::vector<int32_t> damtx;                             // [terms][m]
    std::vector<int32_t> model_slots;
    std::vector<int> keep;                                  // active columns of the last sub-stage that survived (0 = intercept)
    Gram gram;                                              // of the last sub-stage's active columns
    int gram_A = 0;
    // the current sub-stage
    std::vector<int32_t> active_slots;
    int A = 0, vm = 0, n_prev = 0, vm_next = -1;
    Gram cur;                                               // its Gram
    std::vector<int32_t> terms_arr;                         // [A][m] incl. the intercept's zero row, or empty: no matrix-free pass
    int terms_pay_from = 0;
    std::map<std::vector<int64_t>, int64_t> term_ids;
    std::vector<SpectrumRef> early;                         // G2 jobs of the first tests, submitted ahead
    std::vector<std::vector<int32_t>> early_keys;
    std::vector<int32_t> predicted_kills;
    bool have_predicted = false;
    // the coming sub-stage
    Ahead ahead;
    bool have_coming = false;
    int coming_stage = 0;
    std::vector<int32_t> coming_indvec;
    struct Forecast {
        std::vector<int32_t> key;                           // slots of the survivors
        SpectrumRef sp;
    };
    std::vector<Forecast> forecasts;
    std::vector<fokl_outcome *> outcomes, retiring;
    int rc_cb = FOKL_OK;                                    // an error inside a callback of the kill-test loop

    Loop(fokl_run *run, fokl_search *search) : r(run), s(search), ops(run->ops), prm(run->prm), m(run->prm.m) {}

    void release(SpectrumRef &sp)
    {
        if (sp.h) fokl_spectrum_release(s, sp.h);
        sp.h = nullptr;
        sp.gram.reset();
    }

    std::vector<int32_t> columns_without(int count, const std::vector<int32_t> &removed) const
    {
        std::vector<int32_t> out;
        for (int c = 0; c < count; ++c)
            if (std::find(removed.begin(), removed.end(), c) == removed.end()) out.push_back(c);
        return out;
    }

    int fetch_block(Ahead &a)
    {
        if (a.pending) {
            const int nc = a.over + a.T + 1;
            a.block.resize((size_t)a.T * nc);
            const int rc = ops.gram_fetch(ops.ctx, a.block.data(), (int64_t)a.T * nc);
            if (rc != FOKL_OK) return rc;
            a.pending = false;
        }
        return FOKL_OK;
    }

    // engine._build_ahead: K1 + K2 of a coming sub-stage: its columns and their Gram rows against [active | new | y]
    int build_ahead(const std::vector<int32_t> &indvec, const std::vector<int32_t> &active, Ahead *out)
    {
        Ahead a;
        int rc = build_ahead_columns(indvec, &a);
        if (rc == FOKL_OK) rc = launch_ahead_gram(active, &a);
        if (rc != FOKL_OK) return rc;
        *out = std::move(a);
        return FOKL_OK;
    }

    // ... in two halves: K1 (the columns) ...
    int build_ahead_columns(const std::vector<int32_t> &indvec, Ahead *a)
    {
        a->vecs = arrangements(indvec);
        a->T = (int)(a->vecs.size() / (size_t)m);
        int rc = r->pool.take(a->T, &a->slots);
        if (rc != FOKL_OK) return rc;
        if ((rc = ops.build_terms(ops.ctx, a->vecs.data(), a->T, a->slots.data())) != FOKL_OK) return rc;
        r->stats[R_TERMS_PHYSICAL] += a->T;
        a->built = true;
        return FOKL_OK;
    }

    // ... and K2 (their Gram rows against [active | new | y])
    int launch_ahead_gram(const std::vector<int32_t> &active, Ahead *a)
    {
        a->over = (int)active.size();
        std::vector<int32_t> cols(active);
        cols.insert(cols.end(), a->slots.begin(), a->slots.end());
        cols.push_back(FOKL_SLOT_Y);
        const int rc = ops.gram_launch(ops.ctx, a->slots.data(), a->T, cols.data(), (int)cols.size(), 0);
        if (rc != FOKL_OK) return rc;
        a->pending = true;
        a->valid = true;
        return FOKL_OK;
    }

    void release_retired()
    {
        for (fokl_outcome *o : retiring) {
            fokl_outcome_release(s, o);
            fokl_outcome_drop(s, o);
        }
        retiring.clear();
    }

    // engine._retire: every outcome but `keep_a` / `keep_b` can no longer be looked at
    void retire(fokl_outcome *keep_a, fokl_outcome *keep_b)
    {
        std::vector<fokl_outcome *> kept;
        for (fokl_outcome *o : outcomes) {
            if (o == keep_a || o == keep_b) {
                if (std::find(kept.begin(), kept.end(), o) == kept.end()) kept.push_back(o);
            } else {
                retiring.push_back(o);
            }
        }
        outcomes = kept;
    }

    // engine._set_active_terms
    int set_active_terms()
    {
        const int rows = (int)(damtx.size() / (size_t)m);
        std::vector<int64_t> ids((size_t)rows + 1);
        ids[0] = (int64_t)term_ids.emplace(std::vector<int64_t>(), (int64_t)term_ids.size()).first->second;
        for (int t = 0; t < rows; ++t) {
            std::vector<int64_t> key(damtx.begin() + (size_t)t * m, damtx.begin() + (size_t)(t + 1) * m);
            ids[(size_t)t + 1] = term_ids.emplace(std::move(key), (int64_t)term_ids.size()).first->second;
        }
        const int rc = fokl_search_set_substage(s, ids.data(), rows + 1);
        if (rc != FOKL_OK) return rc;
        terms_arr.clear();
        if (prm.matrix_free && rows > 0) {
            const int pay = matrix_free_pay_from(ops, damtx, rows, m);
            if (pay >= 0) {
                terms_arr.assign((size_t)m, 0);
                terms_arr.insert(terms_arr.end(), damtx.begin(), damtx.end());
                terms_pay_from = pay;
            }
        }
        return FOKL_OK;
    }

    int speculate(const std::vector<std::pair<int, bool>> &sizes)
    {
        std::vector<int32_t> sz, mod;
        for (auto &p : sizes) {
            sz.push_back(p.first);
            mod.push_back(p.second ? 1 : 0);
        }
        return fokl_search_speculate(s, sz.data(), mod.data(), (int)sz.size());
    }

    // engine._guess_first_tests (native branch): G2 jobs of the first kill tests the model's least-squares fit makes likely
    // and the tapes of all of them -- across the sub-stage boundary too
    int guess_first_tests(fokl_spectrum *spectrum, int n_new, double siglik, bool before_model,
                          std::vector<std::pair<int, bool>> *tests_out = nullptr)
    {
        std::vector<int32_t> cols((size_t)std::max(1, n_new)), accepted((size_t)std::max(1, n_new));
        int count = 0;
        int rc = fokl_search_likely_first_tests(s, spectrum, n_new, siglik, cols.data(), accepted.data(), &count);
        if (rc != FOKL_OK) return rc;
        std::vector<int32_t> curset;
        std::vector<std::pair<int, bool>> sizes;
        if (before_model) sizes.push_back({A, true});
        fokl_spectrum *against = spectrum;
        (void)fokl_search_hold_spectral(s, 1);
        for (int i = 0; i < count && rc == FOKL_OK; ++i) {
            std::vector<int32_t> trial(curset);
            trial.push_back(cols[(size_t)i]);
            std::sort(trial.begin(), trial.end());
            fokl_spectrum *job = nullptr;
            if ((int)early.size() <= std::min(prm.lookahead_native, 3)) {
                int pos = -1;
                if (against) {
                    const auto without = columns_without(A, curset);
                    pos = (int)(std::lower_bound(without.begin(), without.end(), cols[(size_t)i]) - without.begin());
                }
                const auto idx = columns_without(A, trial);
                rc = fokl_search_spectral_from(s, cur->data(), A + 1, idx.data(), (int)idx.size(), against, pos, &job);
                if (rc != FOKL_OK) break;
                early.push_back({job, cur});
                early_keys.push_back(trial);
            }
            sizes.push_back({A - (int)curset.size() - 1, false});
            if (accepted[(size_t)i]) {
                curset = trial;
                against = job;                              // (NULL: its G2 was not submitted: the next one is decomposed)
            }
        }
        (void)fokl_search_hold_spectral(s, 0);
        if (rc != FOKL_OK) return rc;
        predicted_kills = curset;
        have_predicted = true;
        if (tests_out) tests_out->assign(sizes.begin() + (before_model ? 1 : 0), sizes.end());    // (the likely tests' sizes)
        if (vm_next >= 0 && prm.speculate_across) {
            const int coming = A - (int)curset.size() + vm_next;
            sizes.push_back({coming, true});
            for (int t = 1; t <= vm_next; ++t) sizes.push_back({coming - t, false});
        }
        return speculate(sizes);
    }

    // G2 of the coming sub-stage's model if the kill tests end with that kill set (at most two guesses)
    void foresee(const int32_t *killed, int count)
    {
        if (rc_cb != FOKL_OK || !ahead.valid || forecasts.size() >= 2) return;
        std::vector<int> keep_pred;
        std::vector<int32_t> key;
        for (int c = 0; c < A; ++c)
            if (std::find(killed, killed + count, (int32_t)c) == killed + count) {
                keep_pred.push_back(c);
                if (c > 0) key.push_back(active_slots[(size_t)c]);
            }
        for (auto &f : forecasts)
            if (f.key == key) return;
        if ((rc_cb = fetch_block(ahead)) != FOKL_OK) return;
        Gram g = extend_gram(*cur, A, keep_pred, ahead.block, ahead.T, ahead.over + ahead.T + 1, keep_pred, ahead.over);
        const int L = (int)keep_pred.size() + ahead.T;
        std::vector<int32_t> idx((size_t)L);
        for (int i = 0; i < L; ++i) idx[(size_t)i] = i;
        fokl_spectrum *job = nullptr;
        if ((rc_cb = fokl_search_spectral_from(s, g->data(), L + 1, idx.data(), L, nullptr, -1, &job)) != FOKL_OK) return;
        rc_cb = fokl_search_register_forecast(s, key.data(), (int)key.size(), job, (*g)[(size_t)L * (L + 1) + L]);
        forecasts.push_back({key, {job, g}});
    }

    static void cb_foresee(void *user, const int32_t *killed, int count) { static_cast<Loop *>(user)->foresee(killed, count); }

    static int cb_residual(void *user, const int32_t *idx, int p1, const double *betahat, double *s1, double *s2)
    {
        Loop *l = static_cast<Loop *>(user);
        std::vector<int32_t> slots((size_t)p1);
        for (int i = 0; i < p1; ++i) slots[(size_t)i] = l->active_slots[(size_t)idx[i]];
        double out[2] = {0, 0};
        const int rc = l->ops.bic_resid(l->ops.ctx, slots.data(), p1, betahat, out, 0);
        *s1 = out[0];
        *s2 = out[1];
        return rc;
    }

    int run();
};

int Loop::run()
{
    fokl_run *R = r;
    double tick = now_s();
    auto lap = [&](int which) {
        const double t = now_s();
        R->stats[which] += t - tick;
        tick = t;
    };
    const int half0 = prm.half0, half1 = (int)std::ceil(prm.draws / 2.0 + 1.0);
    // seed Gram: n, sum y, y'y
    gram = std::make_shared<std::vector<double>>(R->base);
    gram_A = 1;
    keep = {0};
    ahead = std::move(R->ahead);
    R->ahead = Ahead();
    int stage = 0;
    std::vector<int32_t> indvec;
    bool have_pattern = R->have_next;
    if (have_pattern) {
        stage = R->next_stage;
        indvec = R->next_indvec;
    } else {
        have_pattern = R->patterns.next(&stage, &indvec);
    }
    std::vector<double> &evs = R->evs;
    fokl_outcome *betas = nullptr, *last = nullptr;
    std::vector<int32_t> mtx, last_damtx;
    int greater = 0;
    int rc = FOKL_OK;

    while (have_pattern) {
        // the pattern after this one (known now: K1 + K2 of it go out during this sub-stage)
        have_coming = R->patterns.next(&coming_stage, &coming_indvec);
        n_prev = 1 + (int)model_slots.size();
        SpectrumRef spectral_job;
        std::vector<int32_t> vecs, new_slots;
        if (ahead.valid) {
            vecs = ahead.vecs;
            new_slots = ahead.slots;
            bool hit = false;
            for (auto &f : forecasts)
                if (!hit && f.key == model_slots) {
                    spectral_job = f.sp;
                    f.sp = SpectrumRef();
                    hit = true;
                }
            if ((rc = fetch_block(ahead)) != FOKL_OK) return rc;
            if (hit) {
                cur = spectral_job.gram;
                R->stats[R_FORECASTS_USED] += 1;
            } else {
                cur = extend_gram(*gram, gram_A, keep, ahead.block, ahead.T, ahead.over + ahead.T + 1, keep, ahead.over);
            }
            fokl_search_clear_forecasts(s);                 // before their Grams go: waits for G2 jobs nobody else holds
            for (auto &f : forecasts) release(f.sp);
            forecasts.clear();
            ahead = Ahead();
        } else {
            // K1 + K2 now: build the new columns, extend the Gram
            vecs = arrangements(indvec);
            const int T = (int)(vecs.size() / (size_t)m);
            if ((rc = R->pool.take(T, &new_slots)) != FOKL_OK) return rc;
            if ((rc = ops.build_terms(ops.ctx, vecs.data(), T, new_slots.data())) != FOKL_OK) return rc;
            R->stats[R_TERMS_PHYSICAL] += T;
            std::vector<int32_t> cols{FOKL_SLOT_ONES};
            cols.insert(cols.end(), model_slots.begin(), model_slots.end());
            cols.insert(cols.end(), new_slots.begin(), new_slots.end());
            cols.push_back(FOKL_SLOT_Y);
            std::vector<double> block((size_t)T * cols.size());
            if ((rc = ops.gram(ops.ctx, new_slots.data(), T, cols.data(), (int)cols.size(), block.data(), 0, 0)) != FOKL_OK)
                return rc;
            std::vector<int> kept((size_t)n_prev);
            for (int i = 0; i < n_prev; ++i) kept[(size_t)i] = i;
            cur = extend_gram(*gram, gram_A, keep, block, T, (int)cols.size(), kept, n_prev);
        }
        vm = (int)(vecs.size() / (size_t)m);
        damtx.insert(damtx.end(), vecs.begin(), vecs.end());
        const int dam = (int)(damtx.size() / (size_t)m);
        if ((rc = set_active_terms()) != FOKL_OK) return rc;
        active_slots.assign(1, FOKL_SLOT_ONES);
        active_slots.insert(active_slots.end(), model_slots.begin(), model_slots.end());
        active_slots.insert(active_slots.end(), new_slots.begin(), new_slots.end());
        A = (int)active_slots.size();
        vm_next = have_coming ? (int)(arrangements(coming_indvec).size() / (size_t)m) : -1;
        early.clear();
        early_keys.clear();
        have_predicted = false;
        bool guessed = false;
        std::vector<std::pair<int, bool>> then;
        if (vm > 0 && A > 1) then.push_back({A - 1, false});
        // G2 of this model may be there already (started while the sub-stage before was being decided): the first kill
        // tests are guessed from it before anything else happens
        if (prm.lookahead > 0 && spectral_job.h && fokl_spectrum_done(spectral_job.h)) {
            if ((rc = guess_first_tests(spectral_job.h, vm, NAN, true, &then)) != FOKL_OK) return rc;
            guessed = true;
        }
        lap(R_PHASE_PREPARE);

        // ---- the sub-stage's model: engine._evaluate ----
        std::vector<int32_t> idx((size_t)A);
        for (int i = 0; i < A; ++i) idx[(size_t)i] = i;
        std::vector<int32_t> then_sizes, then_model;
        for (auto &p : then) {
            then_sizes.push_back(p.first);
            then_model.push_back(p.second ? 1 : 0);
        }
        fokl_spectrum *spectrum = nullptr;
        fokl_tape *tape = nullptr;
        if ((rc = fokl_search_model_begin(s, cur->data(), A + 1, idx.data(), A, spectral_job.h, then_sizes.data(),
                                          then_model.data(), (int)then_sizes.size(), &spectrum, &tape)) != FOKL_OK)
            return rc;
        // K1 of the coming sub-stage: launched while this thread would only wait for the model's eigenpairs (the model's tape
        // is on request, its G2 on its way)
        if (g_build_at_start && have_coming && prm.foresight > 0) {
            if ((rc = build_ahead_columns(coming_indvec, &ahead)) != FOKL_OK) {
                fokl_spectrum_release(s, spectrum);
                return rc;
            }
        }
        const double *buffer = nullptr;
        int p1_check = 0;
        if ((rc = fokl_spectrum_wait(s, spectrum, &buffer, &p1_check)) != FOKL_OK) {
            fokl_spectrum_release(s, spectrum);
            return rc;
        }
        const double *betahat = buffer + 2 * (size_t)A;
        if (!terms_arr.empty() && A >= terms_pay_from) {
            rc = ops.bic_resid_terms_launch(ops.ctx, terms_arr.data() + m, A - 1, betahat);
            R->stats[R_RESID_MATRIX_FREE] += 1;
        } else {
            rc = ops.bic_resid_launch(ops.ctx, active_slots.data(), A, betahat);
        }
        // (the coming sub-stage's columns went out at this sub-stage's start: their Gram rows right behind the residual pass)
        if (rc == FOKL_OK && ahead.built && !ahead.valid) rc = launch_ahead_gram(active_slots, &ahead);
        if (rc == FOKL_OK && !guessed && prm.lookahead > 0 && vm > 0) {
            // G2 of the model has just arrived: the likely first tests' G2 jobs and tapes go out now, under the device's
            // residual pass, not after it
            rc = guess_first_tests(spectrum, vm, NAN, false);
            guessed = rc == FOKL_OK;
        }
        fokl_outcome *full = nullptr;
        if (rc == FOKL_OK) rc = fokl_search_model_commit(s, spectrum, tape, (*cur)[(size_t)A * (A + 1) + A], vm, &full);
        if (rc != FOKL_OK) {
            fokl_spectrum_release(s, spectrum);
            return rc;
        }
        release(spectral_job);
        outcomes.push_back(full);
        // the coming model's G2 for the kill set this model's least-squares fit predicts: its Gram block was launched at the
        // sub-stage's start and lies in front of the residual pass in the device's queue
        auto forecast_now = [&]() {
            if (!(g_build_at_start && prm.forecast_early && ahead.valid && ahead.pending && have_predicted && ops.gram_ready &&
                  forecasts.empty()))
                return;
            if (!ops.gram_ready(ops.ctx)) return;
            foresee(predicted_kills.data(), (int)predicted_kills.size());
            if (rc_cb == FOKL_OK) R->stats[R_FORECASTS_EARLY] += 1;
        };
        forecast_now();
        if (rc_cb != FOKL_OK) return rc_cb;
        {
            const double t0 = now_s();
            double mom[2] = {0, 0};
            if ((rc = ops.bic_resid_fetch(ops.ctx, mom, 0)) != FOKL_OK) return rc;
            R->stats[R_T_RESID] += now_s() - t0;
            forecast_now();
            if (rc_cb != FOKL_OK) return rc_cb;
            double ev_model = 0;
            if ((rc = fokl_search_score(s, full, mom[0], mom[1], n_prev, 0, &ev_model)) != FOKL_OK) return rc;
        }
        fokl_outcome *best = full;
        fokl_outcome_view view{};
        if ((rc = fokl_outcome_info(s, full, &view)) != FOKL_OK) return rc;
        double ev = view.ev;
        lap(R_PHASE_MODEL);

        if (!guessed && prm.lookahead > 0) {
            fokl_spectrum *own = nullptr;
            if ((rc = fokl_outcome_spectrum(s, full, &own)) != FOKL_OK) return rc;
            rc = guess_first_tests(own, vm, view.siglik, false);
            fokl_spectrum_release(s, own);
            if (rc != FOKL_OK) return rc;
        }
        // K1 + K2 of the coming sub-stage now, while this thread would only wait for the model's chain
        release_retired();
        if (have_coming && prm.foresight > 0 && !ahead.valid && !ahead.built) {
            if ((rc = build_ahead(coming_indvec, active_slots, &ahead)) != FOKL_OK) return rc;
        }
        // ... and G2 of the coming model for the kill set the least-squares downdate predicts, as soon as the coming columns'
        // Gram block has arrived -- if the model's chain is there first, the tests go ahead and say it exactly
        if (prm.forecast_early && ahead.valid && ahead.pending && have_predicted && ops.gram_ready && forecasts.empty()) {
            for (int polls = prm.forecast_polls; polls > 0 && !fokl_outcome_chain_ready(full); --polls)
                if (ops.gram_ready(ops.ctx)) {
                    foresee(predicted_kills.data(), (int)predicted_kills.size());
                    if (rc_cb != FOKL_OK) return rc_cb;
                    R->stats[R_FORECASTS_EARLY] += 1;
                    break;
                }
        }
        have_predicted = false;

        // ---- statistics of the new terms (FR:1656-1664) ----
        std::vector<int32_t> new_cols((size_t)vm);
        for (int c = 0; c < vm; ++c) new_cols[(size_t)c] = dam - vm + 1 + c;
        std::vector<double> mean_abs((size_t)std::max(1, vm)), rel_std((size_t)std::max(1, vm));
        if (vm > 0 &&
            (rc = fokl_outcome_new_term_stats(s, full, new_cols.data(), vm, half0, half1, mean_abs.data(), rel_std.data())) != FOKL_OK)
            return rc;
        lap(R_PHASE_STATISTICS);
        R->stat_sizes.push_back(vm);
        R->stat_mean_abs.insert(R->stat_mean_abs.end(), mean_abs.begin(), mean_abs.begin() + vm);
        R->stat_rel_std.insert(R->stat_rel_std.end(), rel_std.begin(), rel_std.begin() + vm);
        std::vector<int> order((size_t)vm);
        for (int c = 0; c < vm; ++c) order[(size_t)c] = c;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return mean_abs[(size_t)x] < mean_abs[(size_t)y]; });
        std::vector<int32_t> cand_col((size_t)std::max(1, vm));
        std::vector<double> mean_sorted((size_t)std::max(1, vm)), rel_sorted((size_t)std::max(1, vm));
        for (int c = 0; c < vm; ++c) {
            cand_col[(size_t)c] = new_cols[(size_t)order[(size_t)c]];
            mean_sorted[(size_t)c] = mean_abs[(size_t)order[(size_t)c]];
            rel_sorted[(size_t)c] = rel_std[(size_t)order[(size_t)c]];
        }

        // ---- the kill tests (FR:1666-1690) ----
        if (prm.update_depth > 6) {
            // The sub-stage after which the stop rule may end the search: the chains of its accepted models are what the
            // search's last act waits for, and those chains wait for G2: short pieces of derived models there
            const bool last_chance = !evs.empty() && greater >= prm.tolerance;
            (void)fokl_search_set_update(s, prm.update_from, last_chance ? 6 : prm.update_depth, prm.update_lookahead);
        }
        std::vector<int32_t> ahead_keys{0}, ahead_offsets{0};
        std::vector<fokl_spectrum *> ahead_spectra;
        ahead_keys.clear();
        for (size_t i = 0; i < early.size(); ++i) {
            ahead_keys.insert(ahead_keys.end(), early_keys[i].begin(), early_keys[i].end());
            ahead_offsets.push_back((int32_t)ahead_keys.size());
            ahead_spectra.push_back(early[i].h);
        }
        if (ahead_keys.empty()) ahead_keys.push_back(0);
        if (ahead_spectra.empty()) ahead_spectra.push_back(nullptr);
        std::vector<int32_t> killed((size_t)std::max(1, vm));
        fokl_kill_tests_args ka{};
        ka.gram = cur->data();
        ka.columns = cand_col.data();
        ka.mean_abs = mean_sorted.data();
        ka.rel_std = rel_sorted.data();
        ka.slots = active_slots.data();
        ka.best = best;
        ka.ahead_keys = ahead_keys.data();
        ka.ahead_offsets = ahead_offsets.data();
        ka.ahead_spectra = ahead_spectra.data();
        ka.user = this;
        ka.foresee = cb_foresee;
        ka.idle_work = nullptr;
        ka.residual = cb_residual;
        ka.active = A;
        ka.proposals = vm;
        ka.n_prev = n_prev;
        ka.vm_next = vm_next;
        ka.ahead_count = (int)early.size();
        fokl_kill_tests_result kr{};
        kr.killed = killed.data();
        rc = fokl_search_kill_tests(s, &ka, &kr);
        for (auto &e : early) release(e);
        early.clear();
        early_keys.clear();
        if (rc == FOKL_OK) rc = rc_cb;
        if (rc != FOKL_OK) return rc;
        if (kr.best_is_new) {
            best = kr.best;
            outcomes.push_back(best);
        }
        ev = kr.evmin;
        killed.resize((size_t)kr.killed_count);
        lap(R_PHASE_TESTS);

        // ---- commit the surviving columns (FR:1691-1695), the stop rule (FR:1701-1721) ----
        keep.clear();
        for (int c = 0; c < A; ++c)
            if (std::find(killed.begin(), killed.end(), (int32_t)c) == killed.end()) keep.push_back(c);
        if (!killed.empty()) {
            std::vector<int32_t> kept_terms, gone;
            for (int t = 0; t < dam; ++t)
                if (std::find(killed.begin(), killed.end(), (int32_t)(t + 1)) == killed.end())
                    kept_terms.insert(kept_terms.end(), damtx.begin() + (size_t)t * m, damtx.begin() + (size_t)(t + 1) * m);
            damtx = std::move(kept_terms);
            for (int32_t c : killed) gone.push_back(active_slots[(size_t)c]);
            R->pool.give(gone);
        }
        model_slots.clear();
        for (size_t i = 1; i < keep.size(); ++i) model_slots.push_back(active_slots[(size_t)keep[i]]);
        gram = cur;
        gram_A = A;
        R->stats[R_SUBSTAGES] += 1;
        last = best;
        last_damtx = damtx;
        bool stop = false;
        if (!evs.empty()) {
            const double lowest = *std::min_element(evs.begin(), evs.end());
            if (ev < lowest) {
                betas = best;
                mtx = damtx;
                greater = 1;
                evs.push_back(ev);
            } else if (greater < prm.tolerance) {
                greater += 1;
                evs.push_back(ev);
            } else {
                evs.push_back(ev);
                stop = true;
            }
        } else {
            greater += 1;
            betas = best;
            mtx = damtx;
            evs.push_back(ev);
        }
        if (stop) {
            lap(R_PHASE_WRAP_UP);
            break;
        }
        retire(betas, best);
        lap(R_PHASE_WRAP_UP);
        have_pattern = have_coming;
        stage = coming_stage;
        indvec = coming_indvec;
    }
    lap(R_PHASE_WRAP_UP);
    // the search stopped: tapes on order for a sub-stage that does not come, the columns built ahead are not needed
    (void)fokl_search_drop_speculation(s);
    fokl_search_clear_forecasts(s);
    for (auto &f : forecasts) release(f.sp);
    forecasts.clear();
    if (ahead.valid) {
        if ((rc = fetch_block(ahead)) != FOKL_OK) return rc;
        R->pool.give(ahead.slots);
        ahead = Ahead();
    }
    if (prm.gimmie) {                                       // FR:1751-1753
        betas = last;
        mtx = last_damtx;
    }
    // the caller takes over the two models that may still be looked at; everything else goes
    for (fokl_outcome *o : outcomes)
        if (o != betas && o != last) retiring.push_back(o);
    outcomes.clear();
    release_retired();
    R->best_model = betas;
    R->last_model = last;
    R->mtx = mtx;
    R->mtx_rows = (int)(mtx.size() / (size_t)m);
    return FOKL_OK;
}

}  // namespace

extern "C" int fokl_run_create(const fokl_backend_ops *ops, const fokl_run_params *params, fokl_run **out)
{
    if (!ops || !params || !out || !ops->reserve_slots || !ops->build_terms || !ops->gram || !ops->gram_launch ||
        !ops->gram_fetch || !ops->bic_resid || !ops->bic_resid_launch || !ops->bic_resid_fetch || params->m < 1 ||
        params->n_phis < 1 || params->draws < 2)
        return run_fail(nullptr, FOKL_ERR_ARG, "fokl_run_create: null pointer, missing entry point or empty problem");
    auto *r = new fokl_run();
    r->ops = *ops;
    r->prm = *params;
    r->pool.ops = &r->ops;
    r->patterns.m = params->m;
    r->patterns.n_phis = params->n_phis;
    r->patterns.way3 = params->way3 != 0;
    r->patterns.sett = params->m == 1 ? 1 : (params->way3 ? 3 : 2);       // FR:1595-1600
    if (params->way3 && params->m == 2) {                   // (FR:1724 reads indvec[2]: the reference raises, the caller's loop too)
        delete r;
        return run_fail(nullptr, FOKL_ERR_ARG, "fokl_run_create: 3-way terms over two inputs");
    }
    int rc = r->pool.grow(std::max(64, params->slot_capacity));
    // the head start: the seed Gram and K1 + K2 of the first sub-stage go to the device now and run while the caller brings
    // its host threads up
    const int32_t seed[2] = {FOKL_SLOT_ONES, FOKL_SLOT_Y};
    r->base.assign(4, 0.0);
    if (rc == FOKL_OK) rc = r->ops.gram(r->ops.ctx, seed, 2, seed, 2, r->base.data(), 0, 0);
    if (rc == FOKL_OK && params->head_start) {
        r->have_next = r->patterns.next(&r->next_stage, &r->next_indvec);
        if (r->have_next) {
            Loop head(r, nullptr);
            rc = head.build_ahead(r->next_indvec, {FOKL_SLOT_ONES}, &r->ahead);
        }
    }
    if (rc != FOKL_OK) {
        delete r;
        return rc;
    }
    *out = r;
    return FOKL_OK;
}

// fokl_search_set_update's arguments as the caller configured its search (known only once the search exists): the loop shortens
// the derivation depth in the sub-stage after which the stop rule may end the search
extern "C" int fokl_run_set_update(fokl_run *r, int from_columns, int depth, int lookahead)
{
    if (!r) return run_fail(nullptr, FOKL_ERR_ARG, "fokl_run_set_update: null run");
    r->prm.update_from = from_columns;
    r->prm.update_depth = depth;
    r->prm.update_lookahead = lookahead;
    return FOKL_OK;
}

extern "C" int fokl_run_search(fokl_run *r, fokl_search *search)
{
    if (!r || !search) return run_fail(r, FOKL_ERR_ARG, "fokl_run_search: null pointer");
    Loop loop(r, search);
    const int rc = loop.run();
    if (rc != FOKL_OK) {
        // what the loop still holds goes back before the caller tears the search down (its handles die with the search)
        for (auto &e : loop.early) loop.release(e);
        for (auto &f : loop.forecasts) loop.release(f.sp);
        for (fokl_outcome *o : loop.outcomes) fokl_outcome_drop(search, o);
        for (fokl_outcome *o : loop.retiring) fokl_outcome_drop(search, o);
        if (loop.ahead.valid && loop.ahead.pending) (void)loop.fetch_block(loop.ahead);
    }
    return rc;
}

extern "C" int fokl_run_result(const fokl_run *r, int32_t *mtx_rows, int32_t *evs_count, int32_t *substages,
                               fokl_outcome **best_model, fokl_outcome **last_model)
{
    if (!r || !mtx_rows || !evs_count || !substages || !best_model || !last_model)
        return run_fail(nullptr, FOKL_ERR_ARG, "fokl_run_result: null pointer");
    *mtx_rows = r->mtx_rows;
    *evs_count = (int32_t)r->evs.size();
    *substages = (int32_t)r->stat_sizes.size();
    *best_model = r->best_model;
    *last_model = r->last_model;
    return FOKL_OK;
}

extern "C" int fokl_run_arrays(const fokl_run *r, int32_t *mtx, double *evs, int32_t *stat_sizes, double *stat_mean_abs,
                               double *stat_rel_std, double *stats)
{
    if (!r) return run_fail(nullptr, FOKL_ERR_ARG, "fokl_run_arrays: null run");
    if (mtx) std::memcpy(mtx, r->mtx.data(), sizeof(int32_t) * r->mtx.size());
    if (evs) std::memcpy(evs, r->evs.data(), sizeof(double) * r->evs.size());
    if (stat_sizes) std::memcpy(stat_sizes, r->stat_sizes.data(), sizeof(int32_t) * r->stat_sizes.size());
    if (stat_mean_abs) std::memcpy(stat_mean_abs, r->stat_mean_abs.data(), sizeof(double) * r->stat_mean_abs.size());
    if (stat_rel_std) std::memcpy(stat_rel_std, r->stat_rel_std.data(), sizeof(double) * r->stat_rel_std.size());
    if (stats) std::memcpy(stats, r->stats, sizeof(double) * FOKL_RUN_STATS);
    return (int)r->stat_mean_abs.size();
}

extern "C" const char *fokl_run_error(const fokl_run *r) { return r ? r->error.c_str() : ""; }

extern "C" void fokl_run_destroy(fokl_run *r)
{
    if (!r) return;
    if (r->ahead.valid && r->ahead.pending) {               // a head start nobody picked up
        r->ahead.block.resize((size_t)r->ahead.T * (size_t)(r->ahead.over + r->ahead.T + 1));
        (void)r->ops.gram_fetch(r->ops.ctx, r->ahead.block.data(), (int64_t)r->ahead.block.size());
    }
    delete r;
}
