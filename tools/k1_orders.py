"""K1 time against the Bernoulli orders in a launch (development aid): T = 56 two-way terms (a, b) over 8 inputs."""
import os, sys
import numpy as np
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from fokl_gpy_amd import _capi, getKernels, engine

ctx = _capi.DeviceContext(0)
n, m = 1_000_000, 8
rng = np.random.default_rng(12)
x = rng.random((n, m)); y = rng.standard_normal(n)
packed, nb, width = getKernels.pack_phis(getKernels.bernoulli(), 1)
ctx.upload(x, y, 1, packed, nb, width)
ctx.reserve_slots(130)
ctx.timing_enable(True)
for pattern in ((1, 0), (2, 0), (4, 0), (8, 0), (2, 1), (3, 1), (3, 2), (4, 1), (4, 3), (5, 4), (8, 7), (12, 11), (20, 19)):
    t = engine.distinct_arrangements(list(pattern) + [0] * 6).astype(np.int32)
    s = np.arange(2, 2 + len(t), dtype=np.int32)
    ctx.build_terms(t, s); ctx.sync(); ctx.timing_reset()
    for _ in range(20):
        ctx.build_terms(t, s)
    ctx.sync()
    tm = ctx.timing_get(_capi.K_BASIS)
    us = tm['ms'] / tm['launches'] * 1e3
    print(f'orders {pattern}: T = {len(t):2d}  {us:6.1f} us  {tm["bytes"] / tm["launches"] / us / 1e3:7.1f} GB/s', flush=True)
