"""
Basis-function coefficient tables ("phis") for the two FoKL kernels.

Mirrors the loader surface of the reference (``src/FoKL/getKernels.py``):

* ``sp500()``      -> tuple[500] of [a, b, c, d] with each an ndarray(499)   (ref GK:221-267)
* ``bernoulli()``  -> tuple[20] of list(n + 2)                                (ref GK:308-326)

The reference's spline table (``splineCoefficient500_highPrecision_smoothed.txt``) is not
distributed with the source tree this project was built against, so ``sp500`` regenerates a
table with the documented recipe (ref GK:270-305 ``bss_anova`` + the cubic-spline fit described in
``docs/_dev/basis_functions``): eigendecompose the BSS-ANOVA kernel kappa_1 on a 500-point grid,
scale each eigenvector by sqrt(eigenvalue), fit a cubic spline through it and express every one of
the 499 pieces in its *local* coordinate t in [0, 1] so that (ref FR:836)

    basis(t) = a + b*t + c*t**2 + d*t**3,        t = 499*x - piece_index    (ref FR:570-589)

The Bernoulli table is the reference's 20x21 coefficient table, stored here as a binary ``.npy``
(``tools/import_bernoulli_table.py`` documents how it was imported).
"""
import os
import functools
import numpy as np

_HERE = os.path.dirname(os.path.realpath(__file__))

N_GRID = 500          # grid points of the kernel matrix == number of spline bases
N_PIECE = N_GRID - 1  # cubic pieces per basis (width 1/499, ref FR:570, FR:584)


def _kappa1(n=N_GRID):
    """BSS-ANOVA first-order kernel on linspace(0,1,n) (Eq. 8 of arXiv:2205.13676; ref GK:280-290)."""
    x = np.linspace(0.0, 1.0, n)
    xi, xj = np.meshgrid(x, x)
    b1 = lambda t: t - 0.5
    b2 = lambda t: t * t - t + 1.0 / 6.0
    b4 = lambda t: t ** 4 - 2.0 * t ** 3 + t * t - 1.0 / 30.0
    return x, b1(xi) * b1(xj) + b2(xi) * b2(xj) - b4(np.abs(xi - xj)) / 24.0


@functools.lru_cache(maxsize=1)
def spline_table(n_basis=N_GRID):
    """
    Dense spline table, ndarray [n_basis, 4, 499] (C-order): table[i, k, p] multiplies t**k on piece p of
    basis i.  Deterministic: eigenvector signs are fixed so that the basis is positive at x = 1 (the
    BSS-ANOVA main-effect convention, cf. B1 = x - 1/2) and bases are ordered by decreasing eigenvalue.
    """
    from scipy.interpolate import CubicSpline
    x, k = _kappa1(N_GRID)
    # LAPACK's result depends (in the last bits) on the BLAS thread count, and the local-coordinate coefficients
    # amplify that by up to 499**3: pin the decomposition to one thread so that the table is reproducible
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1, user_api='blas'):
            lam, vec = np.linalg.eigh(k)
    except ImportError:
        lam, vec = np.linalg.eigh(k)
    order = np.argsort(lam)[::-1]
    lam = lam[order]
    vec = vec[:, order]
    sgn = np.where(vec[-1, :] < 0.0, -1.0, 1.0)
    fun = vec * sgn * np.sqrt(np.abs(lam))            # [500 grid, 500 bases]
    cs = CubicSpline(x, fun[:, :n_basis], axis=0)     # cs.c[j, p, i] * (x - x_p)**(3 - j)
    h = 1.0 / N_PIECE
    tab = np.empty((n_basis, 4, N_PIECE), dtype=np.float64)
    tab[:, 0, :] = cs.c[3].T
    tab[:, 1, :] = (cs.c[2] * h).T
    tab[:, 2, :] = (cs.c[1] * h * h).T
    tab[:, 3, :] = (cs.c[0] * h * h * h).T
    return tab


def table_to_phis(tab):
    """[nb, 4, 499] ndarray -> the reference's tuple-of-lists-of-ndarray format (ref GK:248-255)."""
    return tuple([np.array(tab[i, 0]), np.array(tab[i, 1]), np.array(tab[i, 2]), np.array(tab[i, 3])]
                 for i in range(tab.shape[0]))


def sp500(**kwargs):
    """Return 'phis' for the 'Cubic Splines' kernel: a [500 x 4 x 499] tuple of lists (ref GK:221-267)."""
    allowed = {'Smooth': 0, 'Save': 0}
    for kw in kwargs:
        if kw not in allowed:
            raise ValueError(f"Unexpected keyword argument: {kw}")
    return table_to_phis(spline_table(N_GRID))


def bernoulli(file='bernoulli_bn_scaled.npy'):
    """Return 'phis' for the 'Bernoulli Polynomials' kernel: tuple[20], basis n has n + 2 coefficients (ref GK:308-326)."""
    path = os.path.join(_HERE, 'kernels', file)
    if path.endswith('.npy'):
        coeffs = np.load(path)
    else:
        coeffs = np.loadtxt(path, delimiter=' ', dtype=np.double)
    return tuple(list(coeffs[n, :(n + 2)]) for n in range(coeffs.shape[0]))


# ---------------------------------------------------------------------------------------------------------
# Packing for the device (the C-ABI takes one dense fp64 table; see include/fokl_hip.h, fokl_upload)
# ---------------------------------------------------------------------------------------------------------

KERNEL_SPLINES = 0
KERNEL_BERNOULLI = 1


def pack_phis(phis, kernel_id):
    """
    Flatten 'phis' to the dense fp64 layout the HIP library stages through LDS.

    splines   -> [n_basis, 4, n_piece]  (k-major inside a basis; one basis' slab is 4*499*8 B = 15.6 KB)
    bernoulli -> [n_basis, n_basis + 1] zero padded, row i holds the i + 2 coefficients of order i + 1

    Returns (packed ndarray C-contiguous, n_basis, n_piece_or_row_len).
    """
    nb = len(phis)
    if kernel_id == KERNEL_SPLINES:
        npiece = len(phis[0][0])
        out = np.empty((nb, 4, npiece), dtype=np.float64)
        for i in range(nb):
            for k in range(4):
                out[i, k, :] = np.asarray(phis[i][k], dtype=np.float64)
        return np.ascontiguousarray(out), nb, npiece
    elif kernel_id == KERNEL_BERNOULLI:
        width = max(len(p) for p in phis)
        out = np.zeros((nb, width), dtype=np.float64)
        for i in range(nb):
            out[i, :len(phis[i])] = np.asarray(phis[i], dtype=np.float64)
        return np.ascontiguousarray(out), nb, width
    raise ValueError("unknown kernel id")
