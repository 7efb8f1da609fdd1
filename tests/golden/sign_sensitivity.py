"""How much of the selected model hangs on LAPACK's eigenvector signs at the BASELINE sizes.

The reference draws beta = mun + sigma Q diag(d^1/2) vec (FR:1525-1528): flipping the sign of an eigenvector flips the sign
of that direction's noise -- another, equally valid realisation of the same posterior.  The kill tests (FR:1656-1690)
hinge on Monte-Carlo statistics of those draws, so the untouched reference (LAPACK's signs, which flip under 1-ulp changes
of XtX) and its sign-canonical twin (largest-magnitude component positive, what this product and its goldens use) can
select different models.  This script (test infrastructure, like the golden generators next to it) runs the ORACLE (oracle/fokl_oracle.py) both ways on the same
data and stream and reports where the two searches part.

    python tests/golden/sign_sensitivity.py [case ...]      cases: cfg1 cfg4u0 cfg4u5 cfg2short (default: all)
    python tests/golden/sign_sensitivity.py --last-bits [case ...]

--last-bits asks the other half of the question: is either search reproducible when XtX changes in its last bits (another
BLAS build, another summation order -- this product's Gram comes from an MFMA kernel, not from dgemm)?  Each mode is run
twice, once on XtX as computed and once on XtX (1 + 2^-52 u), u symmetric and uniform in [-1, 1]; a search that parts from
itself under that noise cannot be matched by ANY implementation that does not share the reference host's BLAS bit for bit.
"""
import os
import sys
import time
import warnings

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from fokl_gpy_amd import FoKLRoutines, getKernels
from oracle import fokl_oracle as O

THREADS = int(os.environ.get('GOLDEN_THREADS', '8'))
CASES = {'cfg1': (1, 0, None, {}), 'cfg4u0': (4, 0, None, {}), 'cfg4u5': (4, 5, None, {}),
         # configs[2] with a tenth of the rows: the oracle's per-element column builder needs hours at N = 1e6
         'cfg2short': (2, 0, 100_000, {})}


def run(config, unit, rows, overrides, eigh):
    x, y, spec = bench.config_workload(config, unit, rows)
    fit_kw = dict(spec['fit'])
    fit_kw.update(overrides)
    if spec['kernel'] == 'Cubic Splines':
        tab = np.load(os.path.join(ROOT, 'tests', 'golden', 'spline_phis.npz'))['table']
        phis, kid = getKernels.table_to_phis(tab), O.KERNEL_SPLINES
    else:
        phis, kid = getKernels.bernoulli(), O.KERNEL_BERNOULLI
        if spec['phis_cap']:
            phis = phis[:spec['phis_cap']]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=spec['kernel'], phis=phis, UserWarnings=False, ConsoleOutput=False)
        model.clean(x, y, _setattr=True)                  # formatting + normalisation only (host code, no device)
        inputs, data = model.trainset()
    trace = []
    np.random.seed(spec['seed_fit'])
    build = lambda xsm, phind, phis_, kernel_, terms: O.build_columns_c(xsm, phind, phis_, kernel_, terms, threads=THREADS)
    betas, mtx, evs = O.fit(np.asarray(inputs, dtype=np.float64), np.asarray(data, dtype=np.float64), phis, kid, eigh=eigh,
                            build=build, trace=trace, **fit_kw)
    return betas, mtx, evs, trace


def last_bits(eigh, seed=1):
    """eigh on a Gram matrix whose entries moved by at most one unit in the last place."""
    rng = np.random.default_rng(seed)

    def noisy(A):
        u = rng.uniform(-1.0, 1.0, A.shape)
        return eigh(A * (1.0 + 2.0 ** -52 * (u + u.T) / 2))
    return noisy


def main_last_bits(names):
    print(f"{'case':10s} {'signs':>10s} {'gibbs calls':>12s} {'first call that differs':>24s} {'same model':>10s} "
          f"{'terms':>11s} {'max |d evs| / |evs|':>20s} {'seconds':>8s}")
    for name in names:
        for label, eigh in (('lapack', O.eigh_reference), ('canonical', O.eigh_canonical)):
            t0 = time.time()
            b0, m0, e0, t_0 = run(*CASES[name], eigh)
            b1, m1, e1, t_1 = run(*CASES[name], last_bits(eigh))
            s0, s1 = [t['cols'] for t in t_0], [t['cols'] for t in t_1]
            first = next((i for i, (a, b) in enumerate(zip(s0, s1)) if a != b), None)
            if first is None and len(s0) != len(s1):
                first = min(len(s0), len(s1))
            same = m0.shape == m1.shape and np.array_equal(m0, m1)
            k = min(len(e0), len(e1))
            dev = float(np.max(np.abs(e0[:k] - e1[:k]) / np.abs(e0[:k]))) if k else float('nan')
            print(f"{name:10s} {label:>10s} {f'{len(s0)} / {len(s1)}':>12s} {str(first):>24s} {str(same):>10s} "
                  f"{f'{m0.shape[0]} / {m1.shape[0]}':>11s} {dev:20.3e} {time.time() - t0:8.0f}", flush=True)


def main():
    if sys.argv[1:2] == ['--last-bits']:
        return main_last_bits(sys.argv[2:] or list(CASES))
    names = sys.argv[1:] or list(CASES)
    print(f"{'case':10s} {'gibbs calls':>12s} {'first call that differs':>24s} {'sub-stages':>11s} {'same model':>10s} "
          f"{'terms':>11s} {'max |d evs| / |evs|':>20s} {'seconds':>8s}")
    for name in names:
        t0 = time.time()
        bc, mc, ec, tc = run(*CASES[name], O.eigh_canonical)
        br, mr, er, tr = run(*CASES[name], O.eigh_reference)
        sizes_c, sizes_r = [t['cols'] for t in tc], [t['cols'] for t in tr]
        first = next((i for i, (a, b) in enumerate(zip(sizes_c, sizes_r)) if a != b), None)
        if first is None and len(sizes_c) != len(sizes_r):
            first = min(len(sizes_c), len(sizes_r))
        same = mc.shape == mr.shape and np.array_equal(mc, mr)
        k = min(len(ec), len(er))
        dev = float(np.max(np.abs(ec[:k] - er[:k]) / np.abs(er[:k]))) if k else float('nan')
        print(f"{name:10s} {f'{len(sizes_c)} / {len(sizes_r)}':>12s} {str(first):>24s} {f'{len(ec)} / {len(er)}':>11s} "
              f"{str(same):>10s} {f'{mc.shape[0]} / {mr.shape[0]}':>11s} {dev:20.3e} {time.time() - t0:8.0f}", flush=True)


if __name__ == '__main__':
    main()
