"""K2 right behind K1 (development aid): the Gram kernel's own time when every launch follows a basis-build launch -- into
columns the block does not read, into the block's own new columns (the search's order: K1 then K2 of the same sub-stage) -- against
back to back.  N = 1e6, M = 8."""
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
ctx.reserve_slots(2 + 400)
t21 = engine.distinct_arrangements([2, 1] + [0] * 6).astype(np.int32)
t43 = engine.distinct_arrangements([4, 3] + [0] * 6).astype(np.int32)
t32 = engine.distinct_arrangements([3, 2] + [0] * 6).astype(np.int32)
slots = np.arange(2, 2 + 168, dtype=np.int32)
ctx.build_terms(np.vstack([t21, t32, t43]), slots)
other = np.arange(2 + 200, 2 + 256, dtype=np.int32)
ctx.sync()
ctx.timing_enable(True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for nr, nc in ((56, 75), (56, 103), (56, 139), (56, 171), (28, 73), (8, 101)):
    rs = slots[:nr]
    model = slots[56:56 + nc - nr - 2]
    cs = np.concatenate([[0], model, rs, [1]]).astype(np.int32)
    terms = t21[:nr]
    line = f'gram {nr:3d} x {nc:3d}:'
    for label, pre in (('back to back', None), ('behind K1 into other columns', other[:nr]), ('behind K1 into its own new columns', rs)):
        ctx.gram(rs, cs); ctx.sync(); ctx.timing_reset()
        for _ in range(reps):
            if pre is not None:
                ctx.build_terms(terms, pre)
            ctx.gram(rs, cs)
        t = ctx.timing_get_gram()
        line += f'  {label}: {1e3 * t["ms"] / t["launches"]:6.1f} us (frac {t["ideal_ms"] / t["ms"]:.2f})'
    print(line, flush=True)
