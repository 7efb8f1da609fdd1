// Host side of the Gibbs sampler (G2/G3 of SURVEY 8(a)): numpy-legacy-compatible random stream and the
// D-iteration chain in the eigenbasis of XtX.  N-independent, latency bound -> stays on the host.
//
// Replaces the Python loop /root/reference/src/FoKL/FoKLRoutines.py:1519-1548.  The random stream is the
// third-party numpy legacy RandomState the reference calls (np.random.normal FR:1527, np.random.gamma
// FR:1541/1547); its published algorithm (MT19937 -> 53-bit double -> polar Gaussian with a one-value
// cache shared by normal and gamma -> Marsaglia-Tsang gamma) is restated here and pinned bit-for-bit
// against numpy itself in tests/test_sampler_host.py.
//
// Throughput structure (the stream is inherently serial, the arithmetic on it is not):
//   * normals are produced in blocks: a branch-free rejection pass collects accepted (x1, x2, r2) triples,
//     a second pass evaluates sqrt(-2 log(r2) / r2) for all of them (independent iterations, so the
//     out-of-order core overlaps the libm calls) -- the values and their order are exactly those of
//     numpy's one-at-a-time loop;
//   * per Gibbs iteration the element-wise part (1 / (lamb + 1/tau2), sqrt, w) runs as a vectorisable loop
//     over the block of normals, the three quadratic forms as a second pass.
//
// Must be compiled with -ffp-contract=off: numpy's baseline build rounds every product separately.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#ifdef FOKL_SAMPLER_WIDE
#include <immintrin.h>
#endif

#include "../../include/fokl_hip_internal.h"
#include "fokl_spin.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip

// The serial random stream is the critical path of a fit, so this file is built twice: once for any x86-64 (with AVX2
// clones of the element-wise loops picked at load time) and once, as fokl_sampler_wide.o, for AVX-512 F/DQ/VL/BW,
// where only the tape recorder is exported (fokl_record_tape_wide) and chosen at run time -- measured on Zen 5:
// 151 -> 120 ns per Gibbs iteration at 60 columns.  Same IEEE operations per element either way: identical numbers.
#define FOKL_INTERNAL extern "C" __attribute__((visibility("hidden")))
#ifndef FOKL_NO_FORCE_INLINE
#define FOKL_HOT inline __attribute__((always_inline))
#else
#define FOKL_HOT inline
#endif
#ifdef FOKL_SAMPLER_WIDE
#define FOKL_CLONES
#else
#define FOKL_CLONES __attribute__((target_clones("avx2", "default")))
#endif

namespace {

constexpr int MT_N = 624, MT_M = 397;

// Element-wise pieces of the generator, written as plain loops over arrays so that the compiler vectorises them
// (an AVX2 clone is selected at load time; IEEE operations per element, so lanes change nothing in the results).
FOKL_CLONES
void words_to_doubles(const uint32_t *__restrict__ k, int count, double *__restrict__ out)
{
    for (int j = 0; j < count; ++j) {                   // tempering + numpy's 53-bit double from two words
        uint32_t wa = k[2 * j], wb = k[2 * j + 1];
        wa ^= (wa >> 11);
        wb ^= (wb >> 11);
        wa ^= (wa << 7) & 0x9d2c5680u;
        wb ^= (wb << 7) & 0x9d2c5680u;
        wa ^= (wa << 15) & 0xefc60000u;
        wb ^= (wb << 15) & 0xefc60000u;
        wa ^= (wa >> 18);
        wb ^= (wb >> 18);
        out[j] = ((double)(int32_t)(wa >> 5) * 67108864.0 + (double)(int32_t)(wb >> 6)) / 9007199254740992.0;
    }
}

// x = 2 d - 1 and x^2 for every double of a block: whichever way later draws pair the doubles up (a gamma's uniform
// shifts the pairing by one), a polar attempt is then one addition away: r2 = s[j] + s[j + 1].
FOKL_CLONES
void polar_coordinates(const double *__restrict__ d, int count, double *__restrict__ x, double *__restrict__ sq)
{
    for (int j = 0; j < count; ++j) {
        const double v = 2.0 * d[j] - 1.0;
        x[j] = v;
        sq[j] = v * v;
    }
}

FOKL_CLONES
void polar_candidates(const double *__restrict__ d, int attempts, double *__restrict__ t1, double *__restrict__ t2,
                      double *__restrict__ tr)
{
    for (int i = 0; i < attempts; ++i) {
        const double x1 = 2.0 * d[2 * i] - 1.0, x2 = 2.0 * d[2 * i + 1] - 1.0;
        t1[i] = x1;
        t2[i] = x2;
        tr[i] = x1 * x1 + x2 * x2;
    }
}

FOKL_CLONES
void polar_finish(const double *__restrict__ lg, const double *__restrict__ x1, const double *__restrict__ x2,
                  const double *__restrict__ r2, int count, double *__restrict__ out)
{
    double f[MT_N / 4];                                  // callers pass count <= MT_N / 4
    for (int i = 0; i < count; ++i) f[i] = std::sqrt(-2.0 * lg[i] / r2[i]);   // contiguous: vectorised div + sqrt
    for (int i = 0; i < count; ++i) {
        out[2 * i] = f[i] * x2[i];
        out[2 * i + 1] = f[i] * x1[i];
    }
}

#ifdef FOKL_SAMPLER_WIDE
// bit a of the index set -> bits 2a, 2a + 1 of the mask (an accepted polar attempt keeps both of its coordinates)
constexpr uint8_t PAIR_MASK[16] = {0x00, 0x03, 0x0c, 0x0f, 0x30, 0x33, 0x3c, 0x3f,
                                   0xc0, 0xc3, 0xcc, 0xcf, 0xf0, 0xf3, 0xfc, 0xff};
#endif

struct LegacyRng {
    uint32_t *key;      // 624 words, caller owned (np.random.get_state()[1])
    int pos;
    int has_gauss;
    double gauss;
    // doubles of the current 624-word block, converted in one vectorised pass: entry j <-> words dbase + 2j, + 1
    int dbase = -1, dcount = 0;
    double dbuf[MT_N / 2], xbuf[MT_N / 2], sbuf[MT_N / 2];
    double t1[MT_N / 4 + 1], t2[MT_N / 4 + 1], tr[MT_N / 4 + 1];

    inline void build_dbuf()
    {
        dbase = pos;
        dcount = (MT_N - pos) / 2;
        words_to_doubles(key + pos, dcount, dbuf);
        polar_coordinates(dbuf, dcount, xbuf, sbuf);
    }

    void refill()
    {
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
        uint32_t *k = key;
        for (int i = 0; i < MT_N - MT_M; ++i) {
            const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
            k[i] = k[i + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAG);
        }
        for (int i = MT_N - MT_M; i < MT_N - 1; ++i) {
            const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
            k[i] = k[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAG);
        }
        const uint32_t y = (k[MT_N - 1] & UPPER) | (k[0] & LOWER);
        k[MT_N - 1] = k[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAG);
        pos = 0;
        dbase = -1;
    }

    static inline uint32_t temper(uint32_t y)
    {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }

    inline uint32_t next32()
    {
        if (pos >= MT_N) refill();
        return temper(key[pos++]);
    }

    static inline double to_double(uint32_t wa, uint32_t wb)
    {
        const int32_t a = (int32_t)(wa >> 5), b = (int32_t)(wb >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }

    FOKL_HOT double next_double()
    {
        for (;;) {
            if (dbase >= 0) {
                const int j = (pos - dbase) >> 1;
                if (j < dcount) {
                    pos += 2;
                    return dbuf[j];
                }
            }
            if (pos >= MT_N) refill();
            if (pos + 1 < MT_N) {
                build_dbuf();
                continue;
            }
            // a single word is left in this block: the double straddles the refill
            const uint32_t wa = next32();
            const uint32_t wb = next32();
            dbase = -1;
            return to_double(wa, wb);
        }
    }

    // `count` accepted polar pairs -> 2 * count normals in numpy's order (f * x2 first, then the value numpy
    // would have cached, f * x1).  x1 / x2 / r2 are scratch of length >= count + 1.
    void polar_pairs(int count, double *out, double *x1s, double *x2s, double *r2s)
    {
        int have = 0;
        while (have < count) {
            int j0 = dbase >= 0 ? (pos - dbase) >> 1 : dcount;
            if (dbase < 0 || dcount - j0 < 2) {
                // block (nearly) exhausted or not converted yet: one attempt through the general path
                const double x1 = 2.0 * next_double() - 1.0;
                const double x2 = 2.0 * next_double() - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                x1s[have] = x1;
                x2s[have] = x2;
                r2s[have] = r2;
                have += (r2 < 1.0) & (r2 != 0.0);
                continue;
            }
            const int need = count - have;
            const int attempts = std::min((dcount - j0) / 2, need + need / 2 + 8);
            polar_candidates(dbuf + j0, attempts, t1, t2, tr);
            int used = attempts;
            for (int i = 0; i < attempts; ++i) {         // branch-free compaction of the accepted attempts
                x1s[have] = t1[i];
                x2s[have] = t2[i];
                r2s[have] = tr[i];
                have += (tr[i] < 1.0) & (tr[i] != 0.0);
                if (have == count) {
                    used = i + 1;
                    break;
                }
            }
            pos += 4 * used;                             // attempts computed beyond `used` were never drawn
        }
        double *lg = t1;                                 // scratch: count <= MT_N / 4 + 1 is not guaranteed -> chunk
        for (int base = 0; base < count; base += MT_N / 4) {
            const int c = std::min(MT_N / 4, count - base);
            for (int i = 0; i < c; ++i) lg[i] = std::log(r2s[base + i]);
            polar_finish(lg, x1s + base, x2s + base, r2s + base, c, out + 2 * base);
        }
    }

    // `count` accepted polar attempts left unfinished: out[2i] = x2, out[2i + 1] = x1 (numpy's order of use), r2[i];
    // the normals are f * out[..] with f = sqrt(-2 log(r2) / r2).  Same stream consumption as polar_pairs.
    void polar_pairs_raw(int count, double *__restrict__ out, double *__restrict__ r2)
    {
        int have = 0;
        while (have < count) {
            int j = dbase >= 0 ? (pos - dbase) >> 1 : dcount;
            if (dbase < 0 || dcount - j < 2) {
                // block (nearly) exhausted or not converted yet: one attempt through the general path
                const double x1 = 2.0 * next_double() - 1.0;
                const double x2 = 2.0 * next_double() - 1.0;
                const double rr = x1 * x1 + x2 * x2;
                out[2 * have] = x2;
                out[2 * have + 1] = x1;
                r2[have] = rr;
                have += (rr < 1.0) & (rr != 0.0);
                continue;
            }
            const int j0 = j, last = dcount - 1;
            const double *__restrict__ x = xbuf, *__restrict__ sq = sbuf;
#ifdef FOKL_SAMPLER_WIDE
            // eight attempts per step: r2 by a de-interleaving add, accepted ones packed with vcompresspd (in
            // registers; full-width stores stay inside the `count` entries this call fills anyway)
            const __m512i even = _mm512_setr_epi64(0, 2, 4, 6, 8, 10, 12, 14);
            const __m512i odd = _mm512_setr_epi64(1, 3, 5, 7, 9, 11, 13, 15);
            const __m512d one = _mm512_set1_pd(1.0), zero = _mm512_setzero_pd();
            while (have < count && j + 16 <= dcount) {
                const __m512d s0 = _mm512_loadu_pd(sq + j), s1 = _mm512_loadu_pd(sq + j + 8);
                const __m512d rr = _mm512_add_pd(_mm512_permutex2var_pd(s0, even, s1),
                                                 _mm512_permutex2var_pd(s0, odd, s1));      // sq[j+2a] + sq[j+2a+1]
                unsigned m = _mm512_cmp_pd_mask(rr, one, _CMP_LT_OQ) & _mm512_cmp_pd_mask(rr, zero, _CMP_NEQ_OQ);
                int used = 16;                              // doubles consumed by this step
                const int need = count - have;
                if (__builtin_popcount(m) >= need) {
                    // the last step of a call: keep the first `need` accepted attempts and stop right after the last of
                    // them -- rejected attempts that follow belong to whoever draws next (a uniform, not necessarily
                    // another polar attempt).  The stores below still write full vectors: see the slack
                    // fokl_noise_tape asks for.
                    m = _pdep_u32((1u << need) - 1u, m);
                    used = 2 * (32 - __builtin_clz(m));
                }
                _mm512_storeu_pd(r2 + have, _mm512_maskz_compress_pd((__mmask8)m, rr));
                // (x2, x1) of attempt a = elements 2a+1, 2a: swap within pairs, keep the pairs of accepted attempts
                const __m512d x0 = _mm512_permute_pd(_mm512_loadu_pd(x + j), 0x55);
                const __m512d x1 = _mm512_permute_pd(_mm512_loadu_pd(x + j + 8), 0x55);
                const unsigned lo = m & 15u, hi = m >> 4;
                const int c0 = __builtin_popcount(lo);
                _mm512_storeu_pd(out + 2 * have, _mm512_maskz_compress_pd((__mmask8)PAIR_MASK[lo], x0));
                _mm512_storeu_pd(out + 2 * (have + c0), _mm512_maskz_compress_pd((__mmask8)PAIR_MASK[hi], x1));
                have += c0 + __builtin_popcount(hi);
                j += used;
            }
#endif
            while (j < last && have < count) {              // branch-free compaction of the accepted attempts
                const double rr = sq[j] + sq[j + 1];
                out[2 * have] = x[j + 1];
                out[2 * have + 1] = x[j];
                r2[have] = rr;
                have += (rr < 1.0) & (rr != 0.0);
                j += 2;
            }
            pos += 2 * (j - j0);
        }
    }

    FOKL_HOT double gauss_draw()
    {
        if (has_gauss) {
            const double t = gauss;
            has_gauss = 0;
            gauss = 0.0;
            return t;
        }
        double f, x1, x2, r2;
        do {
            const int j = dbase >= 0 ? (pos - dbase) >> 1 : dcount;
            if (dbase >= 0 && dcount - j >= 2) {            // both doubles in the converted block: coordinates are there
                x1 = xbuf[j];
                x2 = xbuf[j + 1];
                r2 = sbuf[j] + sbuf[j + 1];
                pos += 4;
            } else {
                x1 = 2.0 * next_double() - 1.0;
                x2 = 2.0 * next_double() - 1.0;
                r2 = x1 * x1 + x2 * x2;
            }
        } while (r2 >= 1.0 || r2 == 0.0);
        f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = 1;
        return f * x2;
    }

    inline double std_exponential() { return -std::log(1.0 - next_double()); }

    double std_gamma(double shape)
    {
        if (shape == 1.0) return std_exponential();
        if (shape == 0.0) return 0.0;
        if (shape < 1.0) {
            for (;;) {
                double U = next_double();
                double V = std_exponential();
                if (U <= 1.0 - shape) {
                    double X = std::pow(U, 1.0 / shape);
                    if (X <= V) return X;
                } else {
                    double Y = -std::log((1 - U) / shape);
                    double X = std::pow(1.0 - shape + shape * Y, 1.0 / shape);
                    if (X <= (V + Y)) return X;
                }
            }
        }
        return marsaglia_tsang(shape - 1.0 / 3.0, 1.0 / std::sqrt(9 * (shape - 1.0 / 3.0)));
    }

    // numpy's shape > 1 branch with its two constants b = shape - 1/3, c = 1 / sqrt(9 b) computed by the caller
    // (once per tape instead of once per draw: the sqrt and the divide sat on the recorder's serial path)
    FOKL_HOT double marsaglia_tsang(const double b, const double c)
    {
        for (;;) {
            double X, V;
            do {
                X = gauss_draw();
                V = 1.0 + c * X;
            } while (V <= 0.0);
            V = V * V * V;
            double U = next_double();
            if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
            if (std::log(U) < 0.5 * X * X + b * (1.0 - V + std::log(V))) return b * V;
        }
    }
};

// Scratch for block generation, reused across calls on a thread.
struct Scratch {
    std::vector<double> vals, x1, x2, r2;
    void reserve(int n_normals)
    {
        const size_t pairs = (size_t)n_normals / 2 + 2;
        if (vals.size() < 2 * pairs) vals.resize(2 * pairs);
        if (x1.size() < pairs + 1) {
            x1.resize(pairs + 1);
            x2.resize(pairs + 1);
            r2.resize(pairs + 1);
        }
    }
};

// n successive gauss_draw() results, produced block-wise; identical values, order and final (cache, position).
void fill_normals(LegacyRng &r, Scratch &s, int n, double *out)
{
    int i = 0;
    if (n > 0 && r.has_gauss) {
        out[0] = r.gauss;
        r.has_gauss = 0;
        r.gauss = 0.0;
        i = 1;
    }
    const int remaining = n - i;
    if (remaining <= 0) return;
    const int pairs = (remaining + 1) / 2;
    s.reserve(2 * pairs);
    r.polar_pairs(pairs, s.vals.data(), s.x1.data(), s.x2.data(), s.r2.data());
    std::memcpy(out + i, s.vals.data(), (size_t)remaining * sizeof(double));
    if (remaining & 1) {
        r.gauss = s.vals[2 * pairs - 1];
        r.has_gauss = 1;
    }
}

// The same n draws with the expensive half deferred: full pairs stay raw (see polar_pairs_raw), values that had to
// be formed anyway (a leading cached value; a trailing half pair whose partner goes to the cache) are final.
// *lead = 1 if out[0] is such a leading final value.  r2 has room for n / 2 + 1 entries.
void fill_normals_raw(LegacyRng &r, int n, double *out, double *r2, int32_t *lead)
{
    int i = 0;
    *lead = 0;
    if (n > 0 && r.has_gauss) {
        out[0] = r.gauss;
        r.has_gauss = 0;
        r.gauss = 0.0;
        *lead = 1;
        i = 1;
    }
    const int remaining = n - i;
    if (remaining <= 0) return;
    r.polar_pairs_raw(remaining / 2, out + i, r2);
    if (remaining & 1) out[n - 1] = r.gauss_draw();      // forms the pair, caches its second value
}

// log of `count` values, four at a time through glibc's vector math library (libmvec: within 1 ulp of libm's scalar log,
// equal to it in 3 of 4 arguments, about 5 x its speed).  The finishing half of the polar method is the largest consumer
// of host CPU in a fit (22.7 M logarithms per benchmark fit) and none of it touches the random STREAM -- positions and
// consumption stay numpy's exactly; the normals it yields differ from numpy's in the last bit at most, which reaches the
// posterior draws at the 1e-16 level (tolerance 1e-9).  FOKL_FINISH_LOG=exact keeps libm's scalar log: numpy's bits.
extern "C" __attribute__((visibility("hidden"))) void fokl_logs_avx2(const double *v, int count, double *out);  // fokl_vlog.cpp

inline bool fast_finish_requested()
{
    const char *env = std::getenv("FOKL_FINISH_LOG");
    if (env && std::strcmp(env, "exact") == 0) return false;
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return avx2;
}

// raw [n] + r2 -> vec [n]: the n normals fill_normals would have produced -- bit for bit with fast == false.
// vec may be raw itself (finishing in place).
void finish_normals(const double *raw, const double *__restrict__ r2, int lead, int n, double *vec,
                    double *__restrict__ f, bool fast = false)
{
    const int pairs = (n - lead) / 2;
    if (lead) vec[0] = raw[0];
    if (fast)
        fokl_logs_avx2(r2, pairs, f);
    else
        for (int j = 0; j < pairs; ++j) f[j] = std::log(r2[j]);          // libm calls, independent iterations
    for (int j = 0; j < pairs; ++j) f[j] = std::sqrt(-2.0 * f[j] / r2[j]);
    const double *x = raw + lead;
    double *v = vec + lead;
    for (int j = 0; j < pairs; ++j) {
        const double a = f[j] * x[2 * j], b = f[j] * x[2 * j + 1];
        v[2 * j] = a;
        v[2 * j + 1] = b;
    }
    if ((n - lead) & 1) vec[n - 1] = raw[n - 1];
}

inline bool bind_rng(LegacyRng &r, uint32_t *key, const int32_t *pos, const int32_t *has_gauss, const double *cache)
{
    if (!key || !pos || !has_gauss || !cache) return false;
    if (*pos < 0 || *pos > MT_N) return false;
    r.key = key;
    r.pos = *pos;
    r.has_gauss = *has_gauss ? 1 : 0;
    r.gauss = *cache;
    return true;
}

inline void release_rng(const LegacyRng &r, int32_t *pos, int32_t *has_gauss, double *cache)
{
    *pos = r.pos;
    *has_gauss = r.has_gauss;
    *cache = r.gauss;
}

thread_local Scratch t_scratch;

// One Gibbs iteration's vector half: w = d * qty + sig * (sqrt(d) * v) with d = 1 / (lamb + 1 / tau2), element-wise,
// and the three quadratic forms sum lamb w^2, sum qty w, sum w^2.  Eight partial sums each (element i goes to lane i mod 8), combined in a fixed order: a dependency chain an
// eighth as long as a plain ascending sum's, and the same numbers from this portable loop and from the AVX2 / AVX-512
// statements of it (fokl_vlog.cpp, the wide build below), which are what runs where the CPU has them: IEEE add / mul /
// div / sqrt per element, no contraction.  FOKL_CHAIN_ISA = base | avx2 | avx512 forces one (tests, A/B runs).
extern "C" __attribute__((visibility("hidden"))) void fokl_chain_vector_avx2(const double *lamb, const double *qty,
                                                                             const double *v, int p1, double inv_tau,
                                                                             double sig, double *w, double *out);
extern "C" __attribute__((visibility("hidden"))) void fokl_chain_vector_wide(const double *lamb, const double *qty,
                                                                             const double *v, int p1, double inv_tau,
                                                                             double sig, double *w, double *out);

void chain_vector_portable(const double *__restrict__ lamb, const double *__restrict__ qty, const double *__restrict__ v,
                           int p1, double inv_tau, double sig, double *__restrict__ w, double *__restrict__ out)
{
    double a_lam[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a_ty[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a_ww[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < p1; ++i) {
        const int l = i & 7;
        const double d = 1.0 / (lamb[i] + inv_tau);
        const double wi = d * qty[i] + sig * (std::sqrt(d) * v[i]);
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

#ifndef FOKL_SAMPLER_WIDE
inline int chain_vector_isa()                              // 0 portable, 1 AVX2, 2 AVX-512
{
    static const int isa = [] {
        const bool avx2 = __builtin_cpu_supports("avx2");
        const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
                          __builtin_cpu_supports("avx512vl");
        const char *want = std::getenv("FOKL_CHAIN_ISA");
        if (want && std::strcmp(want, "base") == 0) return 0;
        if (want && std::strcmp(want, "avx2") == 0) return avx2 ? 1 : 0;
        if (want && std::strcmp(want, "avx512") == 0) return wide ? 2 : (avx2 ? 1 : 0);
        return wide ? 2 : (avx2 ? 1 : 0);
    }();
    return isa;
}

inline void chain_vector_part(const double *lamb, const double *qty, const double *v, int p1, double inv_tau, double sig,
                              double *w, double *out)
{
    switch (chain_vector_isa()) {
    case 2: fokl_chain_vector_wide(lamb, qty, v, p1, inv_tau, sig, w, out); break;
    case 1: fokl_chain_vector_avx2(lamb, qty, v, p1, inv_tau, sig, w, out); break;
    default: chain_vector_portable(lamb, qty, v, p1, inv_tau, sig, w, out);
    }
}
#endif

// fokl_noise_tape's loop (see there).  Kept a function of its own with the two gamma draws inlined into the loop
// (FOKL_HOT): with marsaglia_tsang called out of line the generator's position and block pointers went through memory
// on every draw -- 42 + 0.93 p1 ns per iteration against 31 + 0.93 p1 this way (Zen 5, profiles/tape_cost_r02.txt).
__attribute__((noinline)) void record_tape(LegacyRng &r, int p1, int draws, double astar, double atau_star,
                                           double *normals_out, double *pair_r2_out, int32_t *lead_out,
                                           double *gam_sig_out, double *gam_tau_out, int32_t *progress)
{
    const size_t half = (size_t)p1 / 2 + 1;
    const bool fast_sig = astar > 1.0, fast_tau = atau_star > 1.0;     // always, for the hyper-parameters in use
    const double b_sig = astar - 1.0 / 3.0, c_sig = 1.0 / std::sqrt(9 * b_sig);
    const double b_tau = atau_star - 1.0 / 3.0, c_tau = 1.0 / std::sqrt(9 * b_tau);
    for (int k0 = 0; k0 < draws; k0 += FOKL_TAPE_BLOCK) {
        const int k1 = std::min(draws, k0 + FOKL_TAPE_BLOCK);
        for (int k = k0; k < k1; ++k) {
            fill_normals_raw(r, p1, normals_out + (size_t)k * p1, pair_r2_out + (size_t)k * half, lead_out + k);
            gam_sig_out[k] = fast_sig ? r.marsaglia_tsang(b_sig, c_sig) : r.std_gamma(astar);
            gam_tau_out[k] = fast_tau ? r.marsaglia_tsang(b_tau, c_tau) : r.std_gamma(atau_star);
        }
        // iterations up to k1 are complete and visible.  Published per block, not per iteration: every store to a
        // line that other cores are polling costs this thread a coherence round trip.
        if (progress) __atomic_store_n(progress, k1, __ATOMIC_RELEASE);
    }
}

}  // namespace

#ifdef FOKL_SAMPLER_WIDE

// chain_vector_portable with one 512-bit register per set of eight lanes (library-internal).
FOKL_INTERNAL void fokl_chain_vector_wide(const double *__restrict__ lamb, const double *__restrict__ qty,
                                          const double *__restrict__ v, int p1, double inv_tau, double sig,
                                          double *__restrict__ w, double *__restrict__ out)
{
    const __m512d it = _mm512_set1_pd(inv_tau), sg = _mm512_set1_pd(sig), one = _mm512_set1_pd(1.0);
    __m512d lam = _mm512_setzero_pd(), ty = lam, sq = lam;
    int i = 0;
    for (; i + 8 <= p1; i += 8) {
        const __m512d l8 = _mm512_loadu_pd(lamb + i), q8 = _mm512_loadu_pd(qty + i);
        const __m512d d = _mm512_div_pd(one, _mm512_add_pd(l8, it));
        const __m512d w8 = _mm512_add_pd(_mm512_mul_pd(d, q8),
                                         _mm512_mul_pd(sg, _mm512_mul_pd(_mm512_sqrt_pd(d), _mm512_loadu_pd(v + i))));
        _mm512_storeu_pd(w + i, w8);
        const __m512d ww = _mm512_mul_pd(w8, w8);
        lam = _mm512_add_pd(lam, _mm512_mul_pd(l8, ww));
        ty = _mm512_add_pd(ty, _mm512_mul_pd(w8, q8));
        sq = _mm512_add_pd(sq, ww);
    }
    double a_lam[8], a_ty[8], a_ww[8];
    _mm512_storeu_pd(a_lam, lam);
    _mm512_storeu_pd(a_ty, ty);
    _mm512_storeu_pd(a_ww, sq);
    for (int l = 0; i < p1; ++i, ++l) {
        const double d = 1.0 / (lamb[i] + inv_tau);
        const double wi = d * qty[i] + sig * (std::sqrt(d) * v[i]);
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

// The recorder of the AVX-512 build (library-internal); arguments were validated by fokl_noise_tape.
FOKL_INTERNAL void fokl_record_tape_wide(int p1, int draws, double astar, double atau_star, uint32_t *mt_key,
                                         int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                                         double *normals_out, double *pair_r2_out, int32_t *lead_out,
                                         double *gam_sig_out, double *gam_tau_out, int32_t *progress)
{
    LegacyRng r;
    bind_rng(r, mt_key, mt_pos, has_gauss, gauss_cache);
    record_tape(r, p1, draws, astar, atau_star, normals_out, pair_r2_out, lead_out, gam_sig_out, gam_tau_out,
                progress);
    release_rng(r, mt_pos, has_gauss, gauss_cache);
}

#else


FOKL_INTERNAL void fokl_record_tape_wide(int p1, int draws, double astar, double atau_star, uint32_t *mt_key,
                                         int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                                         double *normals_out, double *pair_r2_out, int32_t *lead_out,
                                         double *gam_sig_out, double *gam_tau_out, int32_t *progress);   // fokl_sampler_wide.o

namespace {

bool use_wide_build()
{
    static const bool wide = [] {
        const char *isa = std::getenv("FOKL_SAMPLER_ISA");            // "base" forces the portable build (tests)
        if (isa && std::strcmp(isa, "base") == 0) return false;
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
               __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") &&
               __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
    }();
    return wide;
}

}  // namespace

extern "C" int fokl_rng_normals(uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                                int64_t n, double *out)
{
    LegacyRng r;
    if (!bind_rng(r, mt_key, mt_pos, has_gauss, gauss_cache) || (n > 0 && !out) || n < 0) {
        fokl_set_global_error("fokl_rng_normals: bad RNG state or output pointer");
        return FOKL_ERR_ARG;
    }
    const int64_t block = 256;
    for (int64_t done = 0; done < n; done += block)
        fill_normals(r, t_scratch, (int)std::min<int64_t>(block, n - done), out + done);
    release_rng(r, mt_pos, has_gauss, gauss_cache);
    return FOKL_OK;
}

extern "C" int fokl_rng_gammas(uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                               double shape, double scale, int64_t n, double *out)
{
    LegacyRng r;
    if (!bind_rng(r, mt_key, mt_pos, has_gauss, gauss_cache) || (n > 0 && !out) || !(shape >= 0.0)) {
        fokl_set_global_error("fokl_rng_gammas: bad RNG state, output pointer or shape < 0");
        return FOKL_ERR_ARG;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = scale * r.std_gamma(shape);
    release_rng(r, mt_pos, has_gauss, gauss_cache);
    return FOKL_OK;
}

extern "C" int fokl_gibbs_chain(const double *lamb, const double *qty, int p1, double astar, double atau_star,
                                double b, double btau, double dtd, double sigsqd0, double tausqd0, int draws,
                                uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                                double *w_out, double *sigs_out, double *taus_out)
{
    LegacyRng r;
    if (!lamb || !qty || !w_out || p1 <= 0 || draws < 0 || !bind_rng(r, mt_key, mt_pos, has_gauss, gauss_cache)) {
        fokl_set_global_error("fokl_gibbs_chain: null pointer, empty model or invalid RNG state");
        return FOKL_ERR_ARG;
    }
    if (!(astar >= 0.0) || !(atau_star >= 0.0)) {
        // numpy raises ValueError("shape < 0") here; keep that a hard error.
        fokl_set_global_error("fokl_gibbs_chain: gamma shape parameter is negative or NaN");
        return FOKL_ERR_NUMERIC;
    }

    Scratch &scratch = t_scratch;
    std::vector<double> vec((size_t)p1);
    double sigsqd = sigsqd0, tausqd = tausqd0;
    for (int k = 0; k < draws; ++k) {
        const double inv_tau = 1.0 / tausqd;
        const double sig = std::sqrt(sigsqd);          // sigsqd ** (1/2), FR:1528
        double *__restrict__ w = w_out + (size_t)k * p1;
        const double *__restrict__ v = vec.data();
        fill_normals(r, scratch, p1, vec.data());      // np.random.normal(0, 1, (p1, 1)), C order
        double q[3];
        chain_vector_part(lamb, qty, v, p1, inv_tau, sig, w, q);
        const double q_lam = q[0], q_ty = q[1], q_ww = q[2];
        const double bstar = b + 0.5 * (q_lam - 2.0 * q_ty + dtd + q_ww / tausqd);
        if (bstar < 0.0) {
            sigsqd = NAN;                              // FR:1538-1539: no gamma draw in this branch
        } else {
            sigsqd = 1.0 / ((1.0 / bstar) * r.std_gamma(astar));
        }
        if (sigs_out) sigs_out[k] = sigsqd;
        const double btau_star = (1.0 / (2.0 * sigsqd)) * q_ww + btau;
        tausqd = 1.0 / ((1.0 / btau_star) * r.std_gamma(atau_star));
        if (taus_out) taus_out[k] = tausqd;
    }
    release_rng(r, mt_pos, has_gauss, gauss_cache);
    return FOKL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The same chain in two halves.  Everything random in FR:1519-1548 is independent of the data: per iteration p1
// standard normals, one standard gamma of shape astar and one of shape atau_star (the data only enters through
// the gamma *scales*), unless bstar < 0 skips a draw (FR:1538-1539) -- impossible for b > 0, reported otherwise.
// fokl_noise_tape advances the stream and records those numbers; fokl_gibbs_chain_from_tape replays the
// arithmetic.  The host driver runs the first on a worker thread while eigh / the device residual pass of the
// same and of later candidates proceed (engine.NoisePipeline).
// ---------------------------------------------------------------------------------------------------------

extern "C" int fokl_noise_tape(int p1, int draws, double astar, double atau_star, uint32_t *mt_key, int32_t *mt_pos,
                               int32_t *has_gauss, double *gauss_cache, double *normals_out, double *pair_r2_out,
                               int32_t *lead_out, double *gam_sig_out, double *gam_tau_out, int32_t *progress)
{
    LegacyRng r;
    if (p1 <= 0 || draws < 0 || !normals_out || !pair_r2_out || !lead_out || !gam_sig_out || !gam_tau_out ||
        !bind_rng(r, mt_key, mt_pos, has_gauss, gauss_cache)) {
        fokl_set_global_error("fokl_noise_tape: null pointer, empty model or invalid RNG state");
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        return FOKL_ERR_ARG;
    }
    if (!(astar >= 0.0) || !(atau_star >= 0.0)) {
        fokl_set_global_error("fokl_noise_tape: gamma shape parameter is negative or NaN");
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        return FOKL_ERR_NUMERIC;
    }
    if (use_wide_build()) {
        fokl_record_tape_wide(p1, draws, astar, atau_star, mt_key, mt_pos, has_gauss, gauss_cache, normals_out,
                              pair_r2_out, lead_out, gam_sig_out, gam_tau_out, progress);
        return FOKL_OK;
    }
    record_tape(r, p1, draws, astar, atau_star, normals_out, pair_r2_out, lead_out, gam_sig_out, gam_tau_out,
                progress);
    release_rng(r, mt_pos, has_gauss, gauss_cache);
    return FOKL_OK;
}

namespace {

// A thread that follows a producer: a producer at work is at most a few microseconds away (spin); one that has not
// started yet may be far away (sleep in short steps rather than burn a core -- ranks may share a tight CPU quota).
inline void follow_wait(int &spins)
{
    if (++spins < fokl_spin_budget(2000)) {
        __builtin_ia32_pause();
    } else {
        std::this_thread::sleep_for(std::chrono::microseconds(10));
    }
}

// One Gibbs iteration given its p1 normals and two standard gammas (FR:1521-1548 in the eigenbasis).
struct ChainState {
    double sigsqd, tausqd;
    int32_t flagged = 0;
};

inline void chain_step(const double *__restrict__ lamb, const double *__restrict__ qty, int p1, double b, double btau,
                       double dtd, const double *vec, double gam_sig, double gam_tau, double *__restrict__ w,
                       ChainState &st)
{
    const double inv_tau = 1.0 / st.tausqd;
    const double sig = std::sqrt(st.sigsqd);
    double q[3];
    chain_vector_part(lamb, qty, vec, p1, inv_tau, sig, w, q);
    const double q_lam = q[0], q_ty = q[1], q_ww = q[2];
    const double bstar = b + 0.5 * (q_lam - 2.0 * q_ty + dtd + q_ww / st.tausqd);
    if (bstar < 0.0) {
        st.flagged = 1;                                // the tape holds a gamma the reference would not have drawn
        st.sigsqd = NAN;
    } else {
        st.sigsqd = 1.0 / ((1.0 / bstar) * gam_sig);
    }
    const double btau_star = (1.0 / (2.0 * st.sigsqd)) * q_ww + btau;
    st.tausqd = 1.0 / ((1.0 / btau_star) * gam_tau);
}

}  // namespace

extern "C" int fokl_gibbs_chain_from_tape(const double *lamb, const double *qty, int p1, double b, double btau,
                                          double dtd, double sigsqd0, double tausqd0, int draws,
                                          const double *normals, const double *pair_r2, const int32_t *lead,
                                          const double *gam_sig, const double *gam_tau,
                                          double *w_out, double *sigs_out, double *taus_out, int32_t *bstar_negative,
                                          const int32_t *progress)
{
    if (!lamb || !qty || !normals || !pair_r2 || !lead || !gam_sig || !gam_tau || !w_out || p1 <= 0 || draws < 0) {
        fokl_set_global_error("fokl_gibbs_chain_from_tape: null pointer or empty model");
        return FOKL_ERR_ARG;
    }
    int32_t ready = progress ? 0 : draws;
    ChainState st{sigsqd0, tausqd0};
    const size_t half = (size_t)p1 / 2 + 1;
    std::vector<double> vec((size_t)p1), fbuf(half);
    const bool fast = fast_finish_requested();
    for (int k = 0; k < draws; ++k) {
        for (int spins = 0; ready <= k;) {             // follow a tape that is still being recorded
            ready = __atomic_load_n(progress, __ATOMIC_ACQUIRE);
            if (ready < 0) {
                fokl_set_global_error("fokl_gibbs_chain_from_tape: the noise tape producer failed");
                return FOKL_ERR_STATE;
            }
            if (ready <= k) follow_wait(spins);
        }
        finish_normals(normals + (size_t)k * p1, pair_r2 + (size_t)k * half, lead[k], p1, vec.data(), fbuf.data(), fast);
        chain_step(lamb, qty, p1, b, btau, dtd, vec.data(), gam_sig[k], gam_tau[k], w_out + (size_t)k * p1, st);
        if (sigs_out) sigs_out[k] = st.sigsqd;
        if (taus_out) taus_out[k] = st.tausqd;
    }
    if (bstar_negative) *bstar_negative = st.flagged;
    return FOKL_OK;
}

extern "C" int fokl_finish_tape_blocks(int p1, int draws, double *normals, const double *pair_r2, const int32_t *lead,
                                       const int32_t *progress, int part, int parts, int block, int32_t *block_done)
{
    if (!normals || !pair_r2 || !lead || !block_done || p1 <= 0 || draws < 0 || parts < 1 || part < 0 ||
        part >= parts || block < 1) {
        fokl_set_global_error("fokl_finish_tape_blocks: null pointer, empty model or bad partition");
        return FOKL_ERR_ARG;
    }
    const size_t half = (size_t)p1 / 2 + 1;
    std::vector<double> fbuf(half);
    const bool fast = fast_finish_requested();
    const int nblocks = (draws + block - 1) / block;
    int32_t ready = progress ? 0 : draws;
    for (int blk = part; blk < nblocks; blk += parts) {
        const int k0 = blk * block, k1 = std::min(draws, k0 + block);
        for (int spins = 0; ready < k1;) {
            ready = __atomic_load_n(progress, __ATOMIC_ACQUIRE);
            if (ready < 0) {
                for (int later = blk; later < nblocks; later += parts)
                    __atomic_store_n(block_done + later, -1, __ATOMIC_RELEASE);
                fokl_set_global_error("fokl_finish_tape_blocks: the noise tape producer failed");
                return FOKL_ERR_STATE;
            }
            if (ready < k1) follow_wait(spins);
        }
        for (int k = k0; k < k1; ++k) {
            double *row = normals + (size_t)k * p1;
            finish_normals(row, pair_r2 + (size_t)k * half, lead[k], p1, row, fbuf.data(), fast);
        }
        __atomic_store_n(block_done + blk, 1, __ATOMIC_RELEASE);
    }
    return FOKL_OK;
}

// Rows k0 .. k1 - 1 of a raw tape completed in place (the pool's expanders: fokl_stream_expand leaves raw pairs).
FOKL_INTERNAL void fokl_finish_tape_rows(int p1, double *normals, const double *pair_r2, const int32_t *lead, int k0, int k1)
{
    const size_t half = (size_t)p1 / 2 + 1;
    std::vector<double> fbuf(half);
    const bool fast = fast_finish_requested();
    for (int k = k0; k < k1; ++k) {
        double *row = normals + (size_t)k * p1;
        finish_normals(row, pair_r2 + (size_t)k * half, lead[k], p1, row, fbuf.data(), fast);
    }
}

// fokl_gibbs_chain_from_tape for a tape whose blocks are materialised by other threads: waits on block_done (as
// fokl_gibbs_chain_from_finished_tape does) and completes each row's normals itself.
FOKL_INTERNAL int fokl_gibbs_chain_from_raw_blocks(const double *lamb, const double *qty, int p1, double b, double btau,
                                                   double dtd, double sigsqd0, double tausqd0, int draws,
                                                   const double *normals, const double *pair_r2, const int32_t *lead,
                                                   const double *gam_sig, const double *gam_tau,
                                                   const int32_t *block_done, int block, double *w_out,
                                                   int32_t *bstar_negative)
{
    ChainState st{sigsqd0, tausqd0};
    const size_t half = (size_t)p1 / 2 + 1;
    std::vector<double> vec((size_t)p1), fbuf(half);
    const bool fast = fast_finish_requested();
    int ready_block = -1;
    for (int k = 0; k < draws; ++k) {
        if (k / block > ready_block) {
            const int blk = k / block;
            for (int spins = 0;;) {
                const int32_t flag = __atomic_load_n(block_done + blk, __ATOMIC_ACQUIRE);
                if (flag > 0) break;
                if (flag < 0) {
                    fokl_set_global_error("chain: the tape producer failed or the tape was sent back");
                    return FOKL_ERR_STATE;
                }
                follow_wait(spins);
            }
            ready_block = blk;
        }
        finish_normals(normals + (size_t)k * p1, pair_r2 + (size_t)k * half, lead[k], p1, vec.data(), fbuf.data(), fast);
        chain_step(lamb, qty, p1, b, btau, dtd, vec.data(), gam_sig[k], gam_tau[k], w_out + (size_t)k * p1, st);
    }
    if (bstar_negative) *bstar_negative = st.flagged;
    return FOKL_OK;
}

extern "C" int fokl_gibbs_chain_from_finished_tape(const double *lamb, const double *qty, int p1, double b,
                                                   double btau, double dtd, double sigsqd0, double tausqd0, int draws,
                                                   const double *normals, const double *gam_sig,
                                                   const double *gam_tau, const int32_t *block_done, int block,
                                                   double *w_out, double *sigs_out, double *taus_out,
                                                   int32_t *bstar_negative)
{
    if (!lamb || !qty || !normals || !gam_sig || !gam_tau || !w_out || p1 <= 0 || draws < 0 ||
        (block_done && block < 1)) {
        fokl_set_global_error("fokl_gibbs_chain_from_finished_tape: null pointer or empty model");
        return FOKL_ERR_ARG;
    }
    ChainState st{sigsqd0, tausqd0};
    int ready_block = -1;
    for (int k = 0; k < draws; ++k) {
        if (block_done && k / block > ready_block) {
            const int blk = k / block;
            for (int spins = 0;;) {
                const int32_t flag = __atomic_load_n(block_done + blk, __ATOMIC_ACQUIRE);
                if (flag > 0) break;
                if (flag < 0) {
                    fokl_set_global_error("fokl_gibbs_chain_from_finished_tape: the tape producer failed");
                    return FOKL_ERR_STATE;
                }
                follow_wait(spins);
            }
            ready_block = blk;
        }
        chain_step(lamb, qty, p1, b, btau, dtd, normals + (size_t)k * p1, gam_sig[k], gam_tau[k],
                   w_out + (size_t)k * p1, st);
        if (sigs_out) sigs_out[k] = st.sigsqd;
        if (taus_out) taus_out[k] = st.tausqd;
    }
    if (bstar_negative) *bstar_negative = st.flagged;
    return FOKL_OK;
}

#endif  // FOKL_SAMPLER_WIDE
