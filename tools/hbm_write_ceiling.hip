// Development aid (round 3): what a pure WRITE stream sustains on this GPU, as a function of
//   * the footprint (MI355X_MICROARCH.md quotes 6.0-6.2 TB/s for plain stores into 75 / 302 MB tables: inside or near
//     the 256 MiB Infinity Cache; K1 writes 450 MB of columns per sub-stage that nobody reads before they have left it),
//   * the store width per lane (dword, dwordx2, dwordx4) and the non-temporal bit,
//   * the contiguous bytes a workgroup writes per column (K1: 256 lanes x 16 B = 4 KiB) and the number of columns (streams)
//     written side by side (K1: T = 8 / 28 / 56),
//   * workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_write_ceiling.hip -o /tmp/hbm_write_ceiling && /tmp/hbm_write_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f1;

template <typename T, bool NT>
__global__ __launch_bounds__(256) void write_stream(T *dst, size_t n, T v)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}

// K1's shape: a tile of 512 rows (256 lanes x 2 rows) is written to `streams` columns, one 16-byte store per lane and column
template <bool NT>
__global__ __launch_bounds__(256) void write_columns(d2 *dst, size_t rows2, int streams)
{
    const d2 v = {1.0, 2.0};
    for (size_t tile = blockIdx.x; tile * 256 < rows2; tile += gridDim.x) {
        const size_t i = tile * 256 + threadIdx.x;
        if (i < rows2)
            for (int k = 0; k < streams; ++k) {
                if (NT) __builtin_nontemporal_store(v * (double)(k + 1), dst + (size_t)k * rows2 + i);
                else dst[(size_t)k * rows2 + i] = v * (double)(k + 1);
            }
    }
}

int main(int argc, char **argv)
{
    const bool rows_sweep = argc > 1;                          // any argument: K1's shape against the number of rows only
    const size_t cap = rows_sweep ? (size_t)24 << 30 : (size_t)8 << 30;
    char *buf;
    if (hipMalloc(&buf, cap) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](double bytes, int reps, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        return bytes * reps / ms / 1e6;
    };
    printf("# footprint sweep: dwordx4 stores, 16 workgroups of 256 per CU; GB/s plain | non-temporal\n");
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024, 2048, 4096}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        const int reps = (int)std::max<size_t>(3, ((size_t)8 << 30) / bytes / 4);
        const d2 v = {1.0, 2.0};
        const double a = timeit((double)bytes, reps, [&] { write_stream<d2, false><<<4096, 256>>>((d2 *)buf, n, v); });
        const double b = timeit((double)bytes, reps, [&] { write_stream<d2, true><<<4096, 256>>>((d2 *)buf, n, v); });
        printf("%6zu MiB   %7.0f | %7.0f\n", mb, a, b);
    }
    printf("# store width at 2 GiB (far outside the Infinity Cache), 16 workgroups per CU; GB/s plain | non-temporal\n");
    {
        const size_t bytes = (size_t)2 << 30;
        const double a1 = timeit((double)bytes, 4, [&] { write_stream<float, false><<<4096, 256>>>((float *)buf, bytes / 4, 1.0f); });
        const double b1 = timeit((double)bytes, 4, [&] { write_stream<float, true><<<4096, 256>>>((float *)buf, bytes / 4, 1.0f); });
        const double a2 = timeit((double)bytes, 4, [&] { write_stream<double, false><<<4096, 256>>>((double *)buf, bytes / 8, 1.0); });
        const double b2 = timeit((double)bytes, 4, [&] { write_stream<double, true><<<4096, 256>>>((double *)buf, bytes / 8, 1.0); });
        const d2 v = {1.0, 2.0};
        const double a4 = timeit((double)bytes, 4, [&] { write_stream<d2, false><<<4096, 256>>>((d2 *)buf, bytes / 16, v); });
        const double b4 = timeit((double)bytes, 4, [&] { write_stream<d2, true><<<4096, 256>>>((d2 *)buf, bytes / 16, v); });
        printf("dword     %7.0f | %7.0f\ndwordx2   %7.0f | %7.0f\ndwordx4   %7.0f | %7.0f\n", a1, b1, a2, b2, a4, b4);
    }
    printf("# workgroups per CU at 2 GiB, dwordx4 non-temporal\n");
    for (int per_cu : {1, 2, 4, 8, 16, 32}) {
        const size_t bytes = (size_t)2 << 30;
        const d2 v = {1.0, 2.0};
        const double b = timeit((double)bytes, 4, [&] { write_stream<d2, true><<<256 * per_cu, 256>>>((d2 *)buf, bytes / 16, v); });
        printf("%3d per CU  %7.0f\n", per_cu, b);
    }
    if (rows_sweep) {
        printf("# K1's shape against the number of rows: T = 56 columns side by side, 5 workgroups per CU, one launch each; GB/s plain | nt\n");
        for (size_t n : {(size_t)1000000, (size_t)4000000, (size_t)10000000, (size_t)20000000, (size_t)50000000}) {
            const size_t rows2 = n / 2;
            const double bytes = 56.0 * rows2 * 16;
            const int reps = n >= 10000000 ? 2 : 6;
            const double a = timeit(bytes, reps, [&] { write_columns<false><<<256 * 5, 256>>>((d2 *)buf, rows2, 56); });
            const double b = timeit(bytes, reps, [&] { write_columns<true><<<256 * 5, 256>>>((d2 *)buf, rows2, 56); });
            printf("N = %9zu (%6.0f MB per launch)  %7.0f | %7.0f\n", n, bytes / 1e6, a, b);
        }
        return 0;
    }
    printf("# K1's shape: N = 1e6 rows (8 MB per column), T columns written side by side, 5 workgroups per CU; GB/s plain | nt\n");
    for (int streams : {1, 8, 28, 56, 112, 224}) {
        const size_t rows2 = 500000;
        const double bytes = (double)streams * rows2 * 16;
        const int reps = streams >= 56 ? 6 : 20;
        const double a = timeit(bytes, reps, [&] { write_columns<false><<<256 * 5, 256>>>((d2 *)buf, rows2, streams); });
        const double b = timeit(bytes, reps, [&] { write_columns<true><<<256 * 5, 256>>>((d2 *)buf, rows2, streams); });
        printf("T = %3d (%5.0f MB)  %7.0f | %7.0f\n", streams, bytes / 1e6, a, b);
    }
    return 0;
}
