// Development bench of the two-phase random stream (csrc/fokl_stream.cpp): the walker following the bulk threads over
// `tapes` tapes of 2000 iterations (the shape of a fit's kill tests), and the walker alone over a stream that is already
// there (warm: what the walk itself costs).  Build and run: see tools/stream_bench.sh.
#include <string>
void fokl_set_global_error(const std::string &) {}
extern "C" void fokl_note_thread_cpu(int) {}            // (fokl_hostpool.cpp in the library: per-kind CPU accounting)
#include "../fokl_gpy_amd/csrc/fokl_stream.cpp"
#include <cstdio>
#include <random>
int main(int argc, char **argv)
{
    const int p1 = argc > 1 ? atoi(argv[1]) : 70;
    const int nt = argc > 2 ? atoi(argv[2]) : 2;
    const int tapes = argc > 3 ? atoi(argv[3]) : 200;
    uint32_t key[624];
    std::mt19937 gen(5);
    for (auto &k : key) k = gen();
    fokl_stream *e;
    fokl_stream_create(key, 624, 0, 0.0, nt, nullptr, 0, &e);
    const int draws = 2000;
    std::vector<fokl_tape_row> rows(draws);
    std::vector<double> gs(draws), gt(draws);
    int32_t prog = 0;
    const double astar = 4 + 1 + 1e6 / 2 + p1 / 2.0, atau = 4 + (p1 - 1) / 2.0;
    for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = now_ns();
        const int64_t w0 = e->walker_wait_ns.load(), b0 = e->bulk_busy_ns.load(), s0 = e->segments_made.load();
        for (int i = 0; i < tapes; ++i) {
            fokl_stream_walk(e, p1, draws, astar, atau, rows.data(), gs.data(), gt.data(), &prog);
            fokl_stream_advance_floor(e);
        }
        const auto t1 = now_ns();
        printf("follow  p1 %3d bulk threads %d: %6.1f ns/iter (waiting %5.1f)  bulk %5.1f us/segment, %lld segments, %lld rollbacks\n",
               p1, nt, (double)(t1 - t0) / (tapes * draws), (double)(e->walker_wait_ns.load() - w0) / (tapes * draws),
               (e->bulk_busy_ns.load() - b0) / 1e3 / (double)(e->segments_made.load() - s0),
               (long long)(e->segments_made.load() - s0), (long long)e->rollbacks.load());
    }
    uint64_t hold;
    fokl_stream_hold(e, &hold);
    fokl_stream_cursor cur;
    fokl_stream_tell(e, &cur);
    for (int rep = 0; rep < 3; ++rep) {
        fokl_stream_seek(e, &cur);
        const auto t0 = now_ns();
        for (int i = 0; i < 20; ++i) fokl_stream_walk(e, p1, draws, astar, atau, rows.data(), gs.data(), gt.data(), &prog);
        const auto t1 = now_ns();
        printf("re-walk p1 %3d: %6.1f ns/iter\n", p1, (double)(t1 - t0) / (20 * draws));
    }
    fokl_stream_release(e, hold);
    fokl_stream_destroy(e);
}
