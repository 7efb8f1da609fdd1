// Host side of the Gibbs sampler (G2/G3 of SURVEY 8(a)): numpy-legacy-compatible random stream and the
// D-iteration chain in the eigenbasis of XtX.  N-independent, latency bound -> stays on the host.
//
// Replaces the Python loop /root/reference/src/FoKL/FoKLRoutines.py:1519-1548.  The random stream is the
// third-party numpy legacy RandomState the reference calls (np.random.normal FR:1527, np.random.gamma
// FR:1541/1547); its published algorithm (MT19937 -> 53-bit double -> polar Gaussian with a one-value
// cache shared by normal and gamma -> Marsaglia-Tsang gamma) is restated here and pinned bit-for-bit
// against numpy itself in tests/test_sampler_host.py.
//
// Must be compiled with -ffp-contract=off: numpy's baseline build rounds every product separately.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/fokl_hip.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip

namespace {

struct LegacyRng {
    uint32_t *key;      // 624 words, caller owned (np.random.get_state()[1])
    int pos;
    int has_gauss;
    double gauss;

    inline void refill()
    {
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
        int kk = 0;
        uint32_t y;
        for (; kk < 624 - 397; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
        }
        for (; kk < 623; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
        pos = 0;
    }

    inline uint32_t next32()
    {
        if (pos >= 624) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }

    inline double next_double()
    {
        int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }

    inline double gauss_draw()
    {
        if (has_gauss) {
            const double t = gauss;
            has_gauss = 0;
            gauss = 0.0;
            return t;
        }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
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
        const double b = shape - 1.0 / 3.0;
        const double c = 1.0 / std::sqrt(9 * b);
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

inline bool bind_rng(LegacyRng &r, uint32_t *key, const int32_t *pos, const int32_t *has_gauss, const double *cache)
{
    if (!key || !pos || !has_gauss || !cache) return false;
    if (*pos < 0 || *pos > 624) return false;
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

}  // namespace

extern "C" int fokl_rng_normals(uint32_t *mt_key, int32_t *mt_pos, int32_t *has_gauss, double *gauss_cache,
                                int64_t n, double *out)
{
    LegacyRng r;
    if (!bind_rng(r, mt_key, mt_pos, has_gauss, gauss_cache) || (n > 0 && !out)) {
        fokl_set_global_error("fokl_rng_normals: bad RNG state or output pointer");
        return FOKL_ERR_ARG;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = r.gauss_draw();
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

    double sigsqd = sigsqd0, tausqd = tausqd0;
    for (int k = 0; k < draws; ++k) {
        const double inv_tau = 1.0 / tausqd;
        const double sig = std::sqrt(sigsqd);          // sigsqd ** (1/2), FR:1528
        double *w = w_out + (size_t)k * p1;
        double q_lam = 0.0, q_ty = 0.0, q_ww = 0.0;
        for (int i = 0; i < p1; ++i) {
            const double d = 1.0 / (lamb[i] + inv_tau);
            const double v = r.gauss_draw();           // np.random.normal(0, 1, (p1, 1)), C order
            const double wi = d * qty[i] + sig * (std::sqrt(d) * v);
            w[i] = wi;
            q_lam += lamb[i] * (wi * wi);
            q_ty += wi * qty[i];
            q_ww += wi * wi;
        }
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
