"""K2 (Gram) timing at N = 1e6 for the block shapes of the benchmark workload (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from fokl_gpy_amd import _capi, getKernels, engine

ctx = _capi.DeviceContext(0)
n, m = int(os.environ.get("K2_N", "1000000")), 8
rng = np.random.default_rng(12)
x = rng.random((n, m)); y = rng.standard_normal(n)
packed, nb, width = getKernels.pack_phis(getKernels.bernoulli(), 1)
ctx.upload(x, y, 1, packed, nb, width)
ctx.reserve_slots(2 + 200)
t21 = engine.distinct_arrangements([2, 1] + [0] * 6).astype(np.int32)
t11 = engine.distinct_arrangements([1, 1] + [0] * 6).astype(np.int32)
t32 = engine.distinct_arrangements([3, 2] + [0] * 6).astype(np.int32)
t43 = engine.distinct_arrangements([4, 3] + [0] * 6).astype(np.int32)
slots = np.arange(2, 2 + 56 + 28 + 56 + 56, dtype=np.int32)
ctx.build_terms(np.vstack([t21, t11, t32, t43]), slots)
ctx.sync()
ctx.timing_enable(True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
paths = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2, 3]
# shapes of the benchmark fit's launches: nr new columns against [ones | model | new | y]
SHAPES = ((8, 10), (28, 38), (8, 40), (56, 58), (56, 80), (56, 98), (56, 112), (56, 128), (56, 142), (56, 176), (28, 120),
          (8, 100), (8, 150))
if os.environ.get('K2_SHAPES'):                               # e.g. K2_SHAPES=56x128,8x150
    SHAPES = tuple(tuple(int(v) for v in item.split('x')) for item in os.environ['K2_SHAPES'].split(','))
for nr, nc in SHAPES:
    rs = slots[:nr]
    model = slots[nr:nr + nc - nr - 2]
    cs = np.concatenate([[0], model, rs, [1]]).astype(np.int32)
    line = f'gram {nr:3d} x {nc:3d}:'
    ref = None
    for path in paths:
        g = ctx.gram(rs, cs, path=path)
        if ref is None:
            ref = g
        same = np.array_equal(g, ref)
        ctx.timing_reset()
        for _ in range(reps):
            ctx.gram(rs, cs, path=path)
        t = ctx.timing_get_gram()
        red = ctx.timing_get(_capi.K_GRAM_REDUCE)
        per = t['ms'] / t['launches']
        line += (f'  path {path}: {per * 1e3:7.1f} us {t["bytes"] / t["launches"] / per / 1e6:7.0f} GB/s '
                 f'{t["flops"] / t["launches"] / per / 1e9:6.2f} TF/s frac {t["ideal_ms"] / t["ms"]:.2f} (+ reduction {1e3 * red["ms"] / max(red["launches"], 1):5.1f} us)'
                 f'{"" if same else " DIFFERENT BITS"}')
    if 2 in paths:
        pl = _capi.gram_plan(rs, cs)
        line += f'   [nt {pl["nt"]} ct {pl["ct"]} R {pl["rows_per_chunk"]} ks {pl["ks"]} depth {pl["depth"]} groups {pl["tiles"].shape[0]}]'
    print(line, flush=True)
