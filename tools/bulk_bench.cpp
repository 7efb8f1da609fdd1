// Development bench of the random stream's bulk phase alone (csrc/fokl_stream.cpp): one thread, one segment after the other --
// the recurrence (token section), tempering + accept flags, the rank tables.  Build: g++ -O3 -std=c++17 -ffp-contract=off
// -fno-math-errno -pthread -w -o /tmp/bulk_bench tools/bulk_bench.cpp
#include <string>
void fokl_set_global_error(const std::string &) {}
extern "C" void fokl_note_thread_cpu(int) {}
#include "../fokl_gpy_amd/csrc/fokl_stream.cpp"
#include <cstdio>
#include <random>
int main(int argc, char **argv)
{
    const int segments = argc > 1 ? atoi(argv[1]) : 200;
    fokl_stream *e = new fokl_stream();
    std::mt19937 gen(5);
    for (auto &k : e->key0) k = gen();
    e->wide = cpu_is_wide();
    void *mem;
    posix_memalign(&mem, 64, sizeof(uint32_t) * (size_t)(MT_N + kSegWords + kSegTail + 64));
    uint32_t *scratch = (uint32_t *)mem;
    std::vector<Segment *> segs;
    for (int i = 0; i < 16; ++i) segs.push_back(take_segment());
    for (int rep = 0; rep < 3; ++rep) {
        int64_t t_rec = 0, t_fin = 0, t_tab = 0;
        for (int i = 0; i < segments; ++i) {
            Segment *seg = segs[i % segs.size()];
            int64_t t0 = now_ns();
            generate_raw(e, scratch, i == 0 && rep == 0 ? 0 : 1 + i);
            int64_t t1 = now_ns();
            if (e->wide) finish_segment_wide(seg, 0, scratch + MT_N); else finish_segment_portable(seg, 0, scratch + MT_N);
            int64_t t2 = now_ns();
            if (e->wide) build_rank_tables(seg);
            int64_t t3 = now_ns();
            t_rec += t1 - t0; t_fin += t2 - t1; t_tab += t3 - t2;
        }
        printf("per segment: recurrence %.1f us, temper + flags (+ tables) %.1f us, of which tables %.1f us\n", t_rec / 1e3 / segments,
               t_fin / 1e3 / segments, t_tab / 1e3 / segments);
    }
}
