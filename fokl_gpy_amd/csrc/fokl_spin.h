// How long the library's host threads poll before they sleep.  Every wait of the host pipeline -- a job's completion, the
// verdict on a tape, a tape's next block, a segment of the stream -- spins for a bounded number of `pause`s first (tens of
// microseconds: the things waited for usually arrive within that) and then sleeps.  Several fits side by side on a host with
// a CPU quota pay for every one of those spins out of the quota the others compute with: FOKL_SPIN scales
// the budgets (default 1; bench.py's throughput workers run with 0.05; 0: sleep at once).
#pragma once
#include <cstdlib>

inline int fokl_spin_budget(int pauses)
{
    static const double scale = [] {
        const char *v = std::getenv("FOKL_SPIN");
        const double s = v ? std::atof(v) : 1.0;
        return s < 0.0 ? 0.0 : (s > 16.0 ? 16.0 : s);
    }();
    return (int)(pauses * scale);
}
