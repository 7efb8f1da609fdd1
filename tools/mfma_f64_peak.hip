// Development aid: what v_mfma_f64_16x16x4_f64 sustains on this GPU with operands in registers (no memory traffic) --
// the ceiling the Gram kernel's MFMA-bound launches are to be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o mfma_f64_peak && ./mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void spin(double *out, int iters, double a0, double b0)
{
    d4 acc[ACC];
#pragma unroll
    for (int j = 0; j < ACC; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < ACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACC>
void run(int waves_per_simd)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;      // 256 threads = 4 waves = one per SIMD
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * blocks);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    spin<ACC><<<blocks, 256>>>(out, 100, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<ACC><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)ACC * iters * 4.0 * blocks;
    printf("CUs %d, %d wave(s) per SIMD, %d independent accumulators: %.1f TFLOP/s  (%.1f cycles per MFMA at %d MHz)\n",
           prop.multiProcessorCount, waves_per_simd, ACC, flops / ms / 1e9,
           ms * 1e-3 * prop.clockRate * 1e3 / ((double)ACC * iters * waves_per_simd), prop.clockRate / 1000);
    hipFree(out);
}

int main()
{
    run<1>(1);
    run<4>(1);
    run<8>(1);
    run<8>(2);
    run<12>(2);
    run<8>(4);
    run<4>(8);
    run<2>(8);
    return 0;
}
