"""K1 after idle gaps (development aid): the same T = 56 launch back to back and after 1 / 5 / 20 ms of device idleness,
to separate the kernel from the clock / power state it finds inside a fit (where the device idles between launches)."""
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
ctx.reserve_slots(200)
ctx.timing_enable(True)
t21 = engine.distinct_arrangements([2, 1] + [0] * 6).astype(np.int32)
for rotate in (False, True):
    for gap_ms in (0, 1, 5, 20):
        ctx.build_terms(t21, np.arange(2, 58, dtype=np.int32)); ctx.sync(); ctx.timing_reset()
        for rep in range(20):
            base = 2 + (56 * (rep % 3) if rotate else 0)              # fresh slots each time, or the same ones
            if gap_ms:
                time.sleep(gap_ms * 1e-3)
            ctx.build_terms(t21, np.arange(base, base + 56, dtype=np.int32)); ctx.sync()
        tm = ctx.timing_get(_capi.K_BASIS)
        us = tm['ms'] / tm['launches'] * 1e3
        print(f'rotate slots {rotate!s:5s} idle {gap_ms:2d} ms: {us:6.1f} us per launch, {tm["bytes"] / tm["launches"] / us / 1e3:7.1f} GB/s', flush=True)
