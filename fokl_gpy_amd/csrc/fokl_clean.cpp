// FoKL.clean's two passes over the dataset on several host threads (FoKLRoutines.py:395, 436-437): the per-column minima /
// maxima and x <- (x - lo) / (hi - lo).  Same IEEE operations per element as the reference's numpy statements -- one
// subtraction, one division, separately rounded (-ffp-contract=off; min / max are exact, NaN-propagating as np.min /
// np.max) -- so the normalised inputs are the reference's bit for bit; what changes is that 64 MB are read twice by eight
// threads instead of five times by one (25 -> 3 ms at N = 1e6, M = 8).  The normalised inputs stay a host array: they are
// part of the class surface (self.inputs, FR:1316).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fokl_hip.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip

namespace {

template <typename F>
void over_rows(int64_t n, int threads, F body)
{
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n / 65536 + 1));
    if (threads == 1) {
        body(0, (int64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    const int64_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const int64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        pool.emplace_back(body, t, lo, hi);
    }
    for (auto &th : pool) th.join();
}

}  // namespace

extern "C" int fokl_column_min_max(const double *x, int64_t n, int m, double *lows, double *highs, int threads)
{
    if (!x || !lows || !highs || n < 1 || m < 1 || m > 4096 || threads < 1) {
        fokl_set_global_error("fokl_column_min_max: null pointer or bad shape");
        return FOKL_ERR_ARG;
    }
    threads = std::min(threads, 64);
    std::vector<double> part((size_t)threads * 2 * m);
    std::vector<char> used((size_t)threads, 0);
    over_rows(n, threads, [&](int t, int64_t lo, int64_t hi) {
        double *mn = part.data() + (size_t)t * 2 * m, *mx = mn + m;
        for (int k = 0; k < m; ++k) mn[k] = mx[k] = x[(size_t)lo * m + k];
        for (int64_t i = lo; i < hi; ++i) {
            const double *row = x + (size_t)i * m;
            for (int k = 0; k < m; ++k) {
                const double v = row[k];
                // np.min / np.max propagate NaN: a NaN, once in, stays (comparisons with it are false)
                mn[k] = (v < mn[k] || v != v) ? v : mn[k];
                mx[k] = (v > mx[k] || v != v) ? v : mx[k];
            }
        }
        used[(size_t)t] = 1;
    });
    bool first = true;
    for (int t = 0; t < threads; ++t) {
        if (!used[(size_t)t]) continue;
        const double *mn = part.data() + (size_t)t * 2 * m, *mx = mn + m;
        for (int k = 0; k < m; ++k) {
            if (first) {
                lows[k] = mn[k];
                highs[k] = mx[k];
            } else {
                if (lows[k] == lows[k]) lows[k] = (mn[k] < lows[k] || mn[k] != mn[k]) ? mn[k] : lows[k];
                if (highs[k] == highs[k]) highs[k] = (mx[k] > highs[k] || mx[k] != mx[k]) ? mx[k] : highs[k];
            }
        }
        first = false;
    }
    return FOKL_OK;
}

extern "C" int fokl_normalize_columns(double *x, int64_t n, int m, const double *lows, const double *spans, int threads)
{
    if (!x || !lows || !spans || n < 1 || m < 1 || threads < 1) {
        fokl_set_global_error("fokl_normalize_columns: null pointer or bad shape");
        return FOKL_ERR_ARG;
    }
    over_rows(n, std::min(threads, 64), [&](int, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            double *row = x + (size_t)i * m;
            for (int k = 0; k < m; ++k) row[k] = (row[k] - lows[k]) / spans[k];
        }
    });
    return FOKL_OK;
}
