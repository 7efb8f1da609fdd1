// Development aid: what v_mfma_f64_16x16x4_f64 sustains on this GPU with operands in registers (no memory traffic) --
// the ceiling the Gram kernel's MFMA-bound launches are to be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o mfma_f64_peak && ./mfma_f64_peak
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// stamps[2 b], stamps[2 b + 1] = shader-clock cycles and 100 MHz ticks workgroup b spent in its loop (diagnostic
// build only: MI355X_MICROARCH.md, DVFS item 6 -- in-kernel clock = cycles / ticks x 100 MHz)
template <int ACC>
__global__ __launch_bounds__(256) void spin(double *out, int iters, double a0, double b0, unsigned long long *stamps)
{
    d4 acc[ACC];
#pragma unroll
    for (int j = 0; j < ACC; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < ACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;                 // (waits for the accumulators: the loop has drained)
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// the other fp64 MFMA shape: four 4 x 4 x 4 blocks per instruction (512 flops, one accumulator register per lane)
template <int ACC>
__global__ __launch_bounds__(256) void spin_4x4(double *out, int iters, double a0, double b0)
{
    double acc[ACC];
#pragma unroll
    for (int j = 0; j < ACC; ++j) acc[j] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < ACC; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the way a Gram tile uses it: 4 row-side x 4 column-side operand fragments, 16 accumulators
__global__ __launch_bounds__(256) void spin_4x4_tile(double *out, int iters, double a0, double b0)
{
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = a0 + threadIdx.x * 1e-9 + i;
        b[i] = b0 + threadIdx.x * 1e-9 - i;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run_4x4_tile(int waves_per_simd)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * blocks);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int warm = 0; warm < 10; ++warm) spin_4x4_tile<<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e0);
    spin_4x4_tile<<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 4 * 4 * 4 * 4 * 16.0 * iters * 4.0 * blocks;
    printf("v_mfma_f64_4x4x4, 4 x 4 operand fragments -> 16 accumulators: %d wave(s) per SIMD: %.1f TFLOP/s\n", waves_per_simd,
           flops / ms / 1e9);
    hipFree(out);
}

// both forms in one instruction stream: does the 4x4x4 form fit into the gaps the 16x16x4 form leaves?
template <int SMALL>
__global__ __launch_bounds__(256) void spin_mixed(double *out, int iters, double a0, double b0)
{
    d4 big[4];
    double small[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) big[j] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 8; ++j) small[j] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            big[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, big[j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < SMALL; ++q)
                small[(j * SMALL + q) % 8] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, small[(j * SMALL + q) % 8], 0, 0, 0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += big[j][0] + big[j][1] + big[j][2] + big[j][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += small[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SMALL>
void run_mixed(int waves_per_simd)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * blocks);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int warm = 0; warm < 10; ++warm) spin_mixed<SMALL><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e0);
    spin_mixed<SMALL><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (2048.0 + 512.0 * SMALL) * 4.0 * iters * 4.0 * blocks;
    printf("one 16x16x4 + %d 4x4x4 alternating, %d wave(s) per SIMD: %.1f TFLOP/s in all\n", SMALL, waves_per_simd, flops / ms / 1e9);
    hipFree(out);
}

// does the 16x16x4 form issue faster when it does not accumulate in place?  MODE 0: D = A B + 0, summed by VALU adds;
// MODE 1: D = A B + C with D and C different registers (ping-pong)
template <int MODE>
__global__ __launch_bounds__(256) void spin_noacc(double *out, int iters, double a0, double b0)
{
    d4 acc[8], other[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = other[j] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
                asm volatile("" : "+v"(a));                   // (not loop invariant)
                const d4 t = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, zero, 0, 0, 0);
                acc[j] += t;
            } else if (MODE == 1) {
                other[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, other[j], 0, 0, 0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + other[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run_noacc(int waves_per_simd)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * blocks);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int warm = 0; warm < 10; ++warm) spin_noacc<MODE><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e0);
    spin_noacc<MODE><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2048.0 * 8.0 * (MODE == 0 ? 1 : 2) * iters * 4.0 * blocks;
    printf("16x16x4, %s, %d wave(s) per SIMD: %.1f TFLOP/s of MFMA work\n",
           MODE == 0 ? "C = 0 and the sum kept by VALU adds" : "D and C in different registers", waves_per_simd, flops / ms / 1e9);
    hipFree(out);
}

template <int ACC>
void run_4x4(int waves_per_simd)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * blocks);
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int warm = 0; warm < 10; ++warm) spin_4x4<ACC><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e0);
    spin_4x4<ACC><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 4 * 4 * 4 * 4 * (double)ACC * iters * 4.0 * blocks;
    printf("v_mfma_f64_4x4x4: %d wave(s) per SIMD, %d independent accumulators: %.1f TFLOP/s\n", waves_per_simd, ACC,
           flops / ms / 1e9);
    hipFree(out);
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
    unsigned long long *stamps;
    hipMalloc(&stamps, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    spin<ACC><<<blocks, 256>>>(out, 100, 1.0, 2.0, nullptr);
    hipDeviceSynchronize();
    for (int warm = 0; warm < 40; ++warm) spin<ACC><<<blocks, 256>>>(out, iters, 1.0, 2.0, nullptr);   // ~ a second of load
    hipEventRecord(e0);
    spin<ACC><<<blocks, 256>>>(out, iters, 1.0, 2.0, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks), cyc(blocks);
    for (int b = 0; b < blocks; ++b) {
        ghz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
        cyc[b] = (double)h[2 * b] / ((double)ACC * iters * waves_per_simd);
    }
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    const double flops = 2.0 * 16 * 16 * 4 * (double)ACC * iters * 4.0 * blocks;
    printf("CUs %d, %d wave(s) per SIMD, %d independent accumulators: %.1f TFLOP/s; in-kernel clock %.2f GHz (median over "
           "workgroups), %.1f shader cycles per MFMA and SIMD while resident (median)\n",
           prop.multiProcessorCount, waves_per_simd, ACC, flops / ms / 1e9, ghz[blocks / 2], cyc[blocks / 2]);
    hipFree(stamps);
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
    run_4x4<8>(1);
    run_4x4<16>(1);
    run_4x4<16>(2);
    run_4x4<8>(4);
    run_4x4_tile(1);
    run_4x4_tile(2);
    run_noacc<0>(2);
    run_noacc<0>(4);
    run_noacc<1>(2);
    run_mixed<1>(2);
    run_mixed<2>(2);
    run_mixed<3>(2);
    run_mixed<4>(2);
    return 0;
}
