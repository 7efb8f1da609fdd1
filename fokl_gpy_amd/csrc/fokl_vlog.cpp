// log of a run of doubles through glibc's vector math library (libmvec, AVX2 entry _ZGVdN4v_log): within 1 ulp of libm's
// scalar log, about 5 x its speed.  Its own translation unit because it is the only host code built with -mavx2 -mfma
// outside the AVX-512 recorder; fokl_sampler.cpp calls it only on CPUs that have both (finish_normals, fast mode).
#include <immintrin.h>

extern "C" __m256d _ZGVdN4v_log(__m256d);

extern "C" __attribute__((visibility("hidden"))) void fokl_logs_avx2(const double *__restrict__ v, int count,
                                                                     double *__restrict__ out)
{
    int j = 0;
    for (; j + 4 <= count; j += 4) _mm256_storeu_pd(out + j, _ZGVdN4v_log(_mm256_loadu_pd(v + j)));
    if (j < count) {
        double in[4] = {1.0, 1.0, 1.0, 1.0}, res[4];
        for (int q = 0; j + q < count; ++q) in[q] = v[j + q];
        _mm256_storeu_pd(res, _ZGVdN4v_log(_mm256_loadu_pd(in)));
        for (int q = 0; j + q < count; ++q) out[j + q] = res[q];
    }
}
