"""BASELINE configs[3] family on one GPU (development aid): N rows, M = 16 inputs, 3-way interactions, stages capped by
phis[:K] (the uncapped search is out of reach for the reference and for any exact-parity restatement: the last
sub-stage of stage 6 alone has 3 360 candidate terms, each kill test an eigen-decomposition of that size)."""
import os, sys, time, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
from fokl_gpy_amd import FoKLRoutines, getKernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 3
draws = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.default_rng(13)
x = rng.random((n, 16))
y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] * x[:, 3] + 0.3 * x[:, 4] ** 2 + 0.5 * x[:, 5] * x[:, 6] + 0.05 * rng.standard_normal(n)
phis = getKernels.bernoulli()[:cap]
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = FoKLRoutines.FoKL(kernel=1, phis=phis, way3=True, burnin=draws, draws=draws, UserWarnings=False, ConsoleOutput=True)
    np.random.seed(3)
    t = time.time()
    be, nn, m = model._prepare_fit(x, y, dict(clean=True))
    t1 = time.time()
    be.ctx.timing_enable(True); be.ctx.timing_reset()
    model._search(be, nn, m)
    dt = time.time() - t1
st = model.fit_stats
print('prepare s', round(t1 - t, 2), 'search s', round(dt, 2), 'terms/s', round(st['terms_logical'] / dt, 1))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
print('selected', model.mtx.shape, 'max cols per evaluation', max(t['cols'] for t in model.fit_trace))
for kid, name in ((0, 'basis_build'), (1, 'gram'), (2, 'resid')):
    t = be.ctx.timing_get(kid)
    if t['launches']:
        print(name, 'launches', t['launches'], 'total ms', round(t['ms'], 2), 'GB/s', round(t['bytes'] / t['ms'] / 1e6, 1),
              'TFLOP/s', round(t['flops'] / t['ms'] / 1e9, 2))
