/*
 * ORACLE (test infrastructure, NOT the product): plain-C restatement of the reference's per-element
 * basis-matrix construction.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off, links libm).
 *
 * Restates, element for element and operation for operation (no FMA contraction, libm pow):
 *   - FoKLRoutines._inputs_to_phind        /root/reference/src/FoKL/FoKLRoutines.py:570-589   (F1)
 *   - FoKLRoutines.evaluate_basis, d = 0   /root/reference/src/FoKL/FoKLRoutines.py:834-843   (F2)
 *   - the X-build triple loop of gibbs()   /root/reference/src/FoKL/FoKLRoutines.py:1446-1485 (F3)
 *
 * The reference evaluates `x ** k` on numpy float64 scalars, i.e. C `pow(x, (double)k)` from the same
 * glibc this file links against, sums Bernoulli monomials with Python's sum() (start 0, ascending k,
 * then `c[0] + total`), and multiplies the per-input factors in ascending input order starting from 1.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define ORACLE_KERNEL_SPLINES   0
#define ORACLE_KERNEL_BERNOULLI 1

/* F1: phind = uint16(ceil(x*l)); phind += (phind == 0); phind -= 1; xsm = l*x - phind   (ref FR:570-589) */
void oracle_inputs_to_phind(const double *x, int64_t count, int l_phis, uint16_t *phind, double *xsm)
{
    for (int64_t i = 0; i < count; ++i) {
        double t = x[i] * (double)l_phis;
        uint16_t p = (uint16_t)(int64_t)ceil(t);   /* numpy float64 -> uint16 cast of an in-range value */
        if (p == 0) p = 1;
        p = (uint16_t)(p - 1);
        phind[i] = p;
        xsm[i] = (double)l_phis * x[i] - (double)p;
    }
}

/* F2 cubic: c0 + c1*x + c2*(x**2) + c3*(x**3), left to right (ref FR:836) */
static double basis_cubic(double c0, double c1, double c2, double c3, double x)
{
    double r = c0 + c1 * x;
    r = r + c2 * pow(x, 2.0);
    r = r + c3 * pow(x, 3.0);
    return r;
}

/* F2 Bernoulli: c[0] + sum(c[k]*(x**k) for k in 1..len-1), sum() starts from 0 (ref FR:843) */
static double basis_bernoulli(const double *c, int len, double x)
{
    double s = 0.0;
    for (int k = 1; k < len; ++k)
        s = s + c[k] * pow(x, (double)k);
    return c[0] + s;
}

/*
 * F3: X[i, j] = prod_{k: terms[j,k] != 0} basis_{terms[j,k]}(xsm[i,k]) for T terms (ref FR:1461-1485).
 *
 *   xsm    [N, M] row-major: spline local coordinate, or the normalised input itself for Bernoulli
 *   phind  [N, M] row-major uint16 piece index (splines) or NULL
 *   phis   splines: [n_basis, 4, n_piece]; Bernoulli: [n_basis, width] zero padded (basis i has i + 2 coefficients)
 *   terms  [T, M] int32, entry = basis order (1-based), 0 = input absent
 *   out    column-major [T][ld] (column j contiguous), ld >= N
 */
void oracle_build_columns(const double *xsm, const uint16_t *phind, int64_t N, int M, int kernel,
                          const double *phis, int n_basis, int width,
                          const int32_t *terms, int T, double *out, int64_t ld)
{
    for (int j = 0; j < T; ++j) {
        const int32_t *term = terms + (size_t)j * M;
        double *col = out + (size_t)j * ld;
        for (int64_t i = 0; i < N; ++i) {
            double phi = 1.0;
            for (int k = 0; k < M; ++k) {
                int num = term[k];
                if (num == 0) continue;
                int nid = num - 1;
                double x = xsm[(size_t)i * M + k];
                double b;
                if (kernel == ORACLE_KERNEL_SPLINES) {
                    const double *slab = phis + (size_t)nid * 4 * width;
                    int p = phind[(size_t)i * M + k];
                    b = basis_cubic(slab[p], slab[width + p], slab[2 * width + p], slab[3 * width + p], x);
                } else {
                    b = basis_bernoulli(phis + (size_t)nid * width, nid + 2, x);
                }
                phi = phi * b;
            }
            col[i] = phi;
        }
    }
    (void)n_basis;
}

/* Plain fp64 Gram block G[a, b] = sum_i A[a][i] * B[b][i] over column-major operands (used for size-independent checks). */
void oracle_gram(const double *A, int na, const double *B, int nb, int64_t N, int64_t ld, double *G)
{
    for (int a = 0; a < na; ++a)
        for (int b = 0; b < nb; ++b) {
            const double *ca = A + (size_t)a * ld, *cb = B + (size_t)b * ld;
            double s = 0.0;
            for (int64_t i = 0; i < N; ++i) s += ca[i] * cb[i];
            G[(size_t)a * nb + b] = s;
        }
}
