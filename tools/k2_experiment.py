"""K2 (Gram) timing at N = 1e6 for the block shapes of the benchmark workload (development aid)."""
import os, sys, time
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
ctx.reserve_slots(2 + 200)
t21 = engine.distinct_arrangements([2, 1] + [0] * 6).astype(np.int32)
t11 = engine.distinct_arrangements([1, 1] + [0] * 6).astype(np.int32)
t32 = engine.distinct_arrangements([3, 2] + [0] * 6).astype(np.int32)
slots = np.arange(2, 2 + 56 + 28 + 56, dtype=np.int32)
ctx.build_terms(np.vstack([t21, t11, t32]), slots)
ctx.sync()
ctx.timing_enable(True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for nr, nc in ((8, 10), (28, 38), (56, 58), (56, 98), (56, 142), (28, 120)):
    rs = slots[:nr]
    cs = np.concatenate([[0], slots[:nc - 2], [1]]).astype(np.int32)
    ctx.gram(rs, cs)
    ctx.timing_reset()
    for _ in range(reps):
        ctx.gram(rs, cs)
    t = ctx.timing_get(_capi.K_GRAM)
    per = t['ms'] / t['launches']
    print(f'gram {nr:3d} x {nc:3d}: {per * 1e3:7.1f} us  {t["bytes"] / t["launches"] / per / 1e6:7.1f} GB/s  '
          f'{t["flops"] / t["launches"] / per / 1e9:6.2f} TFLOP/s', flush=True)
