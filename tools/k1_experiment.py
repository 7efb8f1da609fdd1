"""K1 timing experiments at N = 1e6, M = 8 (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from fokl_gpy_amd import _capi, getKernels, engine

ctx = _capi.DeviceContext(0)
n, m = 1_000_000, 8
rng = np.random.default_rng(12)
x = rng.random((n, m)); y = rng.standard_normal(n)
for kid, name in ((1, 'bernoulli'), (0, 'splines')):
    phis = getKernels.bernoulli() if kid == 1 else getKernels.sp500()[:24]
    packed, nb, width = getKernels.pack_phis(phis, kid)
    ctx.upload(x, y, kid, packed, nb, width)
    ctx.reserve_slots(130)
    ctx.timing_enable(True)

    def timeit(term_groups, label, reps=10):
        slots0 = 2
        groups = []
        for t in term_groups:
            groups.append((t.astype(np.int32), np.arange(slots0, slots0 + len(t), dtype=np.int32)))
            slots0 += len(t)
        for t, s in groups: ctx.build_terms(t, s)
        ctx.sync(); ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            for t, s in groups: ctx.build_terms(t, s)
        ctx.sync(); wall = (time.perf_counter() - t0) / reps
        tm = ctx.timing_get(_capi.K_BASIS)
        tot_ms = tm['ms'] / reps
        gbs = tm['bytes'] / reps / (tot_ms * 1e-3) / 1e9
        print(f'{name:9s} {label:34s} launches/iter {tm["launches"]//reps:2d}  device {tot_ms*1e3:7.1f} us  wall {wall*1e6:7.1f} us  {gbs:7.1f} GB/s alg', flush=True)

    t11 = engine.distinct_arrangements([1, 1] + [0] * 6)
    t21 = engine.distinct_arrangements([2, 1] + [0] * 6)
    t1 = engine.distinct_arrangements([1] + [0] * 7)
    t32 = engine.distinct_arrangements([3, 2] + [0] * 6)
    timeit([t1], 'T=8 (1)')
    timeit([t11], 'T=28 (1,1) U=8')
    timeit([t21], 'T=56 (2,1) U=16 one launch')
    timeit([t21[:28], t21[28:]], 'T=56 (2,1) as 2x28')
    timeit([t21[:14], t21[14:28], t21[28:42], t21[42:]], 'T=56 (2,1) as 4x14')
    timeit([t32], 'T=56 (3,2) U=16 one launch')
    # same 56 columns but only 8 distinct factors: terms (1,1) twice
    timeit([np.vstack([t11, t11])], 'T=56 (1,1)x2 U=8 one launch')
