// AVX2 pieces of the host sampler (the only host code built with -mavx2 -mfma outside the AVX-512 recorder; called only
// on CPUs that have both).
//
// log of a run of doubles through glibc's vector math library (libmvec, AVX2 entry _ZGVdN4v_log): within 1 ulp of libm's
// scalar log, about 5 x its speed (finish_normals, fast mode).
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

// One Gibbs iteration's vector half (fokl_sampler.cpp: chain_vector_part is the portable statement of the same
// operations): w = d * qty + sig * (sqrt(d) * v) with d = 1 / (lamb + inv_tau), and the three quadratic forms
// sum lamb w^2, sum qty w, sum w^2 in eight partial sums each (element i -> lane i mod 8) combined in a fixed order.
// IEEE add / mul / div / sqrt per element and no contraction: the same numbers as the portable loop, bit for bit.
extern "C" __attribute__((visibility("hidden"))) void fokl_chain_vector_avx2(
    const double *__restrict__ lamb, const double *__restrict__ qty, const double *__restrict__ v, int p1,
    double inv_tau, double sig, double *__restrict__ w, double *__restrict__ out)
{
    const __m256d it = _mm256_set1_pd(inv_tau), sg = _mm256_set1_pd(sig), one = _mm256_set1_pd(1.0);
    __m256d lam_a = _mm256_setzero_pd(), lam_b = lam_a, ty_a = lam_a, ty_b = lam_a, ww_a = lam_a, ww_b = lam_a;
    int i = 0;
    for (; i + 8 <= p1; i += 8) {
        const __m256d la = _mm256_loadu_pd(lamb + i), lb = _mm256_loadu_pd(lamb + i + 4);
        const __m256d qa = _mm256_loadu_pd(qty + i), qb = _mm256_loadu_pd(qty + i + 4);
        const __m256d da = _mm256_div_pd(one, _mm256_add_pd(la, it)), db = _mm256_div_pd(one, _mm256_add_pd(lb, it));
        const __m256d wa = _mm256_add_pd(_mm256_mul_pd(da, qa),
                                         _mm256_mul_pd(sg, _mm256_mul_pd(_mm256_sqrt_pd(da), _mm256_loadu_pd(v + i))));
        const __m256d wb = _mm256_add_pd(_mm256_mul_pd(db, qb),
                                         _mm256_mul_pd(sg, _mm256_mul_pd(_mm256_sqrt_pd(db), _mm256_loadu_pd(v + i + 4))));
        _mm256_storeu_pd(w + i, wa);
        _mm256_storeu_pd(w + i + 4, wb);
        const __m256d sa = _mm256_mul_pd(wa, wa), sb = _mm256_mul_pd(wb, wb);
        lam_a = _mm256_add_pd(lam_a, _mm256_mul_pd(la, sa));
        lam_b = _mm256_add_pd(lam_b, _mm256_mul_pd(lb, sb));
        ty_a = _mm256_add_pd(ty_a, _mm256_mul_pd(wa, qa));
        ty_b = _mm256_add_pd(ty_b, _mm256_mul_pd(wb, qb));
        ww_a = _mm256_add_pd(ww_a, sa);
        ww_b = _mm256_add_pd(ww_b, sb);
    }
    double a_lam[8], a_ty[8], a_ww[8];
    _mm256_storeu_pd(a_lam, lam_a);
    _mm256_storeu_pd(a_lam + 4, lam_b);
    _mm256_storeu_pd(a_ty, ty_a);
    _mm256_storeu_pd(a_ty + 4, ty_b);
    _mm256_storeu_pd(a_ww, ww_a);
    _mm256_storeu_pd(a_ww + 4, ww_b);
    for (int l = 0; i < p1; ++i, ++l) {
        const double d = 1.0 / (lamb[i] + inv_tau);
        const double wi = d * qty[i] + sig * (__builtin_sqrt(d) * v[i]);
        w[i] = wi;
        const double ww = wi * wi;
        a_lam[l] += lamb[i] * ww;
        a_ty[l] += wi * qty[i];
        a_ww[l] += ww;
    }
    out[0] = ((a_lam[0] + a_lam[4]) + (a_lam[2] + a_lam[6])) + ((a_lam[1] + a_lam[5]) + (a_lam[3] + a_lam[7]));
    out[1] = ((a_ty[0] + a_ty[4]) + (a_ty[2] + a_ty[6])) + ((a_ty[1] + a_ty[5]) + (a_ty[3] + a_ty[7]));
    out[2] = ((a_ww[0] + a_ww[4]) + (a_ww[2] + a_ww[6])) + ((a_ww[1] + a_ww[5]) + (a_ww[3] + a_ww[7]));
}
