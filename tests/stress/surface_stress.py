"""Random fitted models: evaluate / coverage3 / bss_derivatives through the device against the CPU checker backend of the
tests (same model object, same draws) -- development aid / stress run.  Tolerance 1e-8 of the output's scale: the device's
Bernoulli columns differ from the libm-pow ones by an ulp of the largest monomial, which order-13 terms turn into 1e-9."""
import os, sys, warnings, copy
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fokl_gpy_amd import FoKLRoutines, getKernels
from helpers import OracleBackend
warnings.simplefilter('ignore')
spl = getKernels.table_to_phis(np.load(os.path.join(ROOT, 'tests', 'golden', 'spline_phis.npz'))['table'])
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(200, 4000)), int(rng.integers(1, 6))
    x = rng.random((n, m)) * rng.uniform(0.5, 20, size=m) - rng.uniform(0, 5, size=m)      # raw, un-normalised inputs
    y = np.sin(x[:, 0]) + (0.1 * x[:, 1 % m] * x[:, 2 % m] if m > 1 else 0) + 0.1 * rng.standard_normal(n)
    kw = dict(burnin=int(rng.integers(20, 80)), draws=int(rng.integers(65, 300)), tolerance=int(rng.integers(1, 3)),
              UserWarnings=False, ConsoleOutput=False)
    kw.update(dict(kernel=1) if rng.integers(0, 2) else dict(kernel=0, phis=spl))
    model = FoKLRoutines.FoKL(**kw)
    np.random.seed(seed)
    model.fit(x, y, clean=True)
    xe = rng.random((int(rng.integers(1, 700)), m)) * (x.max(0) - x.min(0)) + x.min(0)
    out = {}
    for name, use_oracle in (('dev', False), ('cpu', True)):
        mdl = copy.copy(model)
        if use_oracle:
            mdl._backend_override = OracleBackend()
        np.random.seed(seed + 1)
        mean, bounds = mdl.evaluate(xe, clean=True, ReturnBounds=True)
        np.random.seed(seed + 1)
        mean_only = mdl.evaluate(xe, clean=True)
        np.random.seed(seed + 1)
        cov = mdl.coverage3()
        lo_hi = np.asarray(mdl.minmax, dtype=float)                      # bss_derivatives takes normalised inputs (FR:594-660)
        xn = (xe - lo_hi[:, 0]) / (lo_hi[:, 1] - lo_hi[:, 0])
        d12 = mdl.bss_derivatives(inputs=xn, d1=True, d2=True)
        dfull = mdl.bss_derivatives(inputs=xn, d1=True, d2=False, ReturnFullArray=True, IndividualDraws=bool(seed % 2), draws=7)
        out[name] = (mean, bounds, mean_only, cov[0], cov[1], np.asarray(cov[2]), np.asarray(d12), np.asarray(dfull))
    ok = True
    for a, b in zip(out['dev'], out['cpu']):
        if a.shape != b.shape or not np.allclose(a, b, rtol=1e-7, atol=1e-8 * (np.abs(b).max() + 1e-300)):
            ok = False
    bad += not ok
    if not ok:
        for k, (a, b) in enumerate(zip(out['dev'], out['cpu'])):
            if a.shape != b.shape or not np.allclose(a, b, rtol=1e-7, atol=1e-8 * (np.abs(b).max() + 1e-300)):
                print('   item', k, a.shape, b.shape, 'max abs diff', (np.abs(a - b).max() if a.shape == b.shape else None), 'scale', np.abs(b).max())
    print(seed, 'rows', n, 'inputs', m, 'kernel', kw['kernel'], 'terms', model.mtx.shape[0], 'OK' if ok else 'MISMATCH ' + str([a.shape == b.shape and bool(np.allclose(a, b, rtol=1e-7, atol=1e-8 * (np.abs(b).max() + 1e-300))) for a, b in zip(out['dev'], out['cpu'])]), flush=True)
print('mismatches', bad)
