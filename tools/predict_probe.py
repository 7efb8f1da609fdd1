import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from fokl_gpy_amd import _capi, getKernels
ctx = _capi.DeviceContext(0)
n = 1_000_000
rng = np.random.default_rng(1)
x = rng.random((n, 8)); y = rng.standard_normal(n)
packed, nb, width = getKernels.pack_phis(getKernels.bernoulli(), 1)
ctx.upload(x, y, 1, packed, nb, width)
ctx.reserve_slots(80)
from fokl_gpy_amd import engine
t = engine.distinct_arrangements([1, 1] + [0] * 6).astype(np.int32)
ctx.build_terms(t, np.arange(2, 30, dtype=np.int32)); ctx.sync()
ctx.timing_enable(True)
for nc, draws in ((38, 1000), (38, 250), (38, 64), (10, 1000), (20, 1000)):
    sl = np.concatenate([[0], np.arange(2, 2 + 28)]).astype(np.int32)
    sl = np.resize(sl, nc).astype(np.int32)
    betas = rng.standard_normal((draws, nc))
    cut = int(np.floor(draws * 0.025) + 1)
    ctx.predict(sl, betas, cut); ctx.timing_reset()
    ctx.predict(sl, betas, cut)
    tm = ctx.timing_get(_capi.K_PREDICT)
    print(f'nc {nc} draws {draws}: {tm["ms"]:.2f} ms  {tm["flops"]/tm["ms"]/1e9:.2f} TFLOP/s', flush=True)
