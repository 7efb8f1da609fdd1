"""K2 stress on integer data (exact answers): random blocks whose last row tile is ragged (1 .. 8 row-side columns of 16,
the case the half-tile slots of gram_tiles_dma_kernel take), random overlap between the row-side and the column-side
lists, random N -- every path against numpy.    python tools/k2_ragged_stress.py [seed] [trials]"""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
from fokl_gpy_amd import _capi, getKernels

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = _capi.DeviceContext(0)
packed, nb, width = getKernels.pack_phis(getKernels.bernoulli(), 1)
bad = halves = 0
for trial in range(trials):
    n = int(rng.choice([1, 31, 33, 500, 1000, 4099, 30011, 120001]))
    C = 260
    ctx.upload(rng.random((n, 1)), rng.integers(-3, 4, n).astype(float), 1, packed, nb, width)
    cols = rng.integers(-3, 4, size=(n, C)).astype(np.float64)
    ctx.reserve_slots(2 + C)
    for j in range(C):
        ctx.write_slot(2 + j, cols[:, j])
    full = np.concatenate([np.ones((n, 1)), ctx.read_slot(1)[:, None], cols], axis=1)     # by slot number
    nr = 16 * int(rng.integers(1, 12)) + int(rng.integers(1, 9))                          # ragged last row tile
    nr = min(nr, 200)
    nc = int(rng.integers(max(3, nr // 3), C))
    rs = (2 + rng.permutation(C)[:nr]).astype(np.int32)
    kind = int(rng.integers(0, 3))
    if kind == 0:                                             # the search's pattern: [ones | others | the row-side columns | y]
        rest = np.setdiff1d(np.arange(2, 2 + C), rs)
        k = max(0, min(nc - nr - 2, rest.shape[0]))
        cs = np.concatenate([[0], rest[:k], rs, [1]]).astype(np.int32)
    elif kind == 1:                                           # partial overlap, random order
        cs = rng.permutation(np.arange(0, 2 + C))[:nc].astype(np.int32)
    else:                                                     # no overlap
        cs = np.setdiff1d(np.arange(0, 2 + C), rs)[:nc].astype(np.int32)
    want = full[:, rs].T @ full[:, cs]
    pl = _capi.gram_plan(rs, cs)
    halves += int(pl['half'].sum())
    for path in ((2, 3) if os.environ.get('K2_STRESS_PANELS') else (2,)):     # path 3: development builds only
        got = ctx.gram(rs, cs, path=path)
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH trial {trial} n {n} block {nr} x {cs.shape[0]} kind {kind} path {path}: "
                  f"{int((got != want).sum())} elements differ", flush=True)
    print(f"{trial} n {n} block {nr} x {cs.shape[0]} kind {kind} half tiles {int(pl['half'].sum())} ok", flush=True)
print('half tiles exercised', halves, 'mismatches', bad)
sys.exit(1 if bad else 0)
