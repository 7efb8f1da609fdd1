// Development aid: what HBM sustains on this GPU for the access mixes of K1 (mostly writes) and K3 (reads): 16-byte
// per-lane streams, grid-stride, over buffers far larger than the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_rw_peak.hip -o hbm_rw_peak && ./hbm_rw_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void write_only(d2 *dst, size_t n, int nt)
{
    const d2 v = {1.0, 2.0};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (nt) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void read_only(const d2 *src, size_t n, double *out)
{
    double s = 0.0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const d2 v = src[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void copy(const d2 *src, d2 *dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(src[i], dst + i);
}
// one read stream feeding seven write streams: the T = 56, 8-input shape of K1
__global__ __launch_bounds__(256) void one_to_seven(const d2 *src, d2 *dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const d2 v = src[i];
#pragma unroll
        for (int k = 0; k < 7; ++k) __builtin_nontemporal_store(v * (double)(k + 1), dst + (size_t)k * n + i);
    }
}

int main()
{
    const size_t n = (size_t)1 << 26;                      // 2^26 d2 = 1 GiB per stream
    d2 *a, *b;
    double *out;
    hipMalloc(&a, n * sizeof(d2));
    hipMalloc(&b, 7 * n * sizeof(d2));
    hipMalloc(&out, 8);
    hipMemset(a, 0, n * sizeof(d2));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](const char *name, double bytes, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f GB/s\n", name, bytes * 5 / ms / 1e6);
    };
    const int grid = 256 * 16;
    timeit("write only (plain stores)", n * 16.0, [&] { write_only<<<grid, 256>>>(b, n, 0); });
    timeit("write only (non-temporal)", n * 16.0, [&] { write_only<<<grid, 256>>>(b, n, 1); });
    timeit("read only", n * 16.0, [&] { read_only<<<grid, 256>>>(a, n, out); });
    timeit("copy (1 read : 1 write)", n * 32.0, [&] { copy<<<grid, 256>>>(a, b, n); });
    timeit("1 read : 7 writes", n * 16.0 * 8, [&] { one_to_seven<<<grid, 256>>>(a, b, n); });
    return 0;
}
