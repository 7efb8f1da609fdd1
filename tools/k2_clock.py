"""Development aid: in-kernel clock of the Gram kernel (diagnostic build with -DFOKL_GT_STAMP, see tools/k2_clock.sh)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from fokl_gpy_amd import _capi, getKernels, engine
ctx = _capi.DeviceContext(0)
n, m = int(os.environ.get('K2_N', '1000000')), 8
rng = np.random.default_rng(12)
x = rng.random((n, m)); y = rng.standard_normal(n)
packed, nb, width = getKernels.pack_phis(getKernels.bernoulli(), 1)
ctx.upload(x, y, 1, packed, nb, width)
ctx.reserve_slots(2 + 200)
terms = np.vstack([engine.distinct_arrangements(p + [0] * 6) for p in ([2, 1], [1, 1], [3, 2], [4, 3])]).astype(np.int32)
slots = np.arange(2, 2 + terms.shape[0], dtype=np.int32)
ctx.build_terms(terms, slots); ctx.sync()
lib = ctypes.CDLL(_capi.LIB_PATH)
SHAPES = ((56, 58), (56, 128), (56, 176), (8, 150))
if os.environ.get('K2_SHAPES'):
    SHAPES = tuple(tuple(int(v) for v in item.split('x')) for item in os.environ['K2_SHAPES'].split(','))
for nr, nc in SHAPES:
    rs = slots[:nr]; cs = np.concatenate([[0], slots[nr:nc - 2], rs, [1]]).astype(np.int32)
    ctx.sync(); assert lib.fokl_debug_stamps_clear() == 0
    ctx.timing_enable(True); ctx.timing_reset()
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):          # a second or so of back-to-back launches
        ctx.gram(rs, cs, path=2)
    tg = ctx.timing_get_gram()
    ctx.timing_enable(False)
    st = np.zeros(8192, dtype=np.uint64)
    assert lib.fokl_debug_stamps_read(st.ctypes.data_as(ctypes.c_void_p), 8192) == 0
    cyc, ticks = st[0::2].astype(float), st[1::2].astype(float)
    ok = ticks > 0
    ghz = cyc[ok] / ticks[ok] * 0.1
    us = np.sort(ticks[ok]) / 100
    print(f'gram {nr} x {nc}: in-kernel clock {np.median(ghz):.2f} GHz (median over {ok.sum()} workgroups; min {ghz.min():.2f}, max {ghz.max():.2f}); '
          f'a workgroup\'s loop takes {np.median(us):.0f} us (5 % {us[int(0.05 * len(us))]:.0f}, 95 % {us[int(0.95 * len(us))]:.0f}, max {us[-1]:.0f}), '
          f'the launch {1e3 * tg["ms"] / tg["launches"]:.0f} us (HIP events)', flush=True)
