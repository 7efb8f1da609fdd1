// Development aid: what does the other hardware thread of the recorder's core have to give?  fokl_noise_tape alone on a
// core, two of them on the two hardware threads of one core, two on different cores (ns per Gibbs iteration each).
//   g++ -O2 -std=c++17 -pthread tools/tape_smt_bench.cpp -o tape_smt_bench -ldl && ./tape_smt_bench fokl_gpy_amd/libfokl_hip.so
#include <sched.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
typedef int (*tape_fn)(int, int, double, double, uint32_t *, int32_t *, int32_t *, double *, double *, double *, int32_t *,
                       double *, double *, int32_t *);
static tape_fn f;

// The same with a fresh set of buffers per tape out of `sets` (sets * ~(12 p + 24) * 2000 bytes: beyond the caches when
// large, as in a fit, where 379 tapes of a few hundred KB each are written per 84 ms).
static double run_cold(int cpu, int p, int reps, int sets)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpu, &set);
    sched_setaffinity(0, sizeof(set), &set);
    const int D = 2000;
    std::vector<uint32_t> key(624);
    for (int i = 0; i < 624; i++) key[i] = i * 2654435761u + 1 + cpu;
    int32_t pos = 624, hg = 0;
    double c = 0;
    struct Bufs { std::vector<double> nm, r2, g1, g2; std::vector<int32_t> lead; };
    std::vector<Bufs> b(sets);
    for (auto &x : b) {
        x.nm.assign((size_t)D * p + 16, 0.0);
        x.r2.assign((size_t)D * (p / 2 + 1) + 8, 0.0);
        x.g1.assign(D, 0.0);
        x.g2.assign(D, 0.0);
        x.lead.assign(D, 0);
    }
    auto t = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) {
        Bufs &x = b[i % sets];
        f(p, D, 5e5, 4.0 + (p - 1) / 2.0, key.data(), &pos, &hg, &c, x.nm.data(), x.r2.data(), x.lead.data(), x.g1.data(), x.g2.data(), nullptr);
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count() / reps / D * 1e9;
}

static double run(int cpu, int p, int reps)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpu, &set);
    sched_setaffinity(0, sizeof(set), &set);
    const int D = 2000;
    std::vector<uint32_t> key(624);
    for (int i = 0; i < 624; i++) key[i] = i * 2654435761u + 1 + cpu;
    int32_t pos = 624, hg = 0;
    double c = 0;
    std::vector<double> nm((size_t)D * p + 16), r2((size_t)D * (p / 2 + 1) + 8), g1(D), g2(D);
    std::vector<int32_t> lead(D);
    for (int i = 0; i < 5; i++) f(p, D, 5e5, 30.0, key.data(), &pos, &hg, &c, nm.data(), r2.data(), lead.data(), g1.data(), g2.data(), nullptr);
    auto t = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) f(p, D, 5e5, 30.0, key.data(), &pos, &hg, &c, nm.data(), r2.data(), lead.data(), g1.data(), g2.data(), nullptr);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count() / reps / D * 1e9;
}

int main(int argc, char **argv)
{
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("%s\n", dlerror()); return 1; }
    f = (tape_fn)dlsym(h, "fokl_noise_tape");
    cpu_set_t mine;
    sched_getaffinity(0, sizeof(mine), &mine);
    int a = -1;
    for (int c = 0; c < CPU_SETSIZE && a < 0; ++c) if (CPU_ISSET(c, &mine)) a = c;
    std::ifstream sib("/sys/devices/system/cpu/cpu" + std::to_string(a) + "/topology/thread_siblings_list");
    std::string line;
    std::getline(sib, line);
    int s = -1;
    for (size_t i = 0, start = 0; i <= line.size(); ++i)
        if (i == line.size() || line[i] == ',' || line[i] == '-') {
            const int v = std::atoi(line.substr(start, i - start).c_str());
            if (v != a && CPU_ISSET(v, &mine)) s = v;
            start = i + 1;
        }
    int other = -1;
    for (int c = a + 1; c < CPU_SETSIZE && other < 0; ++c) if (CPU_ISSET(c, &mine) && c != s) other = c;
    printf("cpu %d, its sibling %d (siblings list '%s'), another core %d\n", a, s, line.c_str(), other);
    if (argc > 2 && std::string(argv[2]) == "cold") {
        for (int p : {2, 10, 19, 28, 38, 47, 60}) {
            const double warm = run_cold(a, p, 400, 1), cold = run_cold(a, p, 400, 200);
            printf("p = %3d (tau shape as in a fit): one buffer set %.1f ns per iteration; 200 sets in rotation %.1f\n", p, warm, cold);
        }
        return 0;
    }
    for (int p : {2, 60, 120}) {
        const double alone = run(a, p, 100);
        double r1 = 0, r2v = 0, r3 = 0, r4 = 0;
        if (s >= 0) {
            std::thread t1([&] { r1 = run(a, p, 100); }), t2([&] { r2v = run(s, p, 100); });
            t1.join();
            t2.join();
        }
        if (other >= 0) {
            std::thread t3([&] { r3 = run(a, p, 100); }), t4([&] { r4 = run(other, p, 100); });
            t3.join();
            t4.join();
        }
        printf("p = %3d: alone %.1f ns per iteration; on the two hardware threads of one core %.1f / %.1f; on two cores %.1f / %.1f\n", p, alone, r1, r2v, r3, r4);
    }
    return 0;
}
