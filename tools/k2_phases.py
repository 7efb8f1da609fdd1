"""Development aid: cycles per phase of a chunk in gram_tiles_dma_kernel (diagnostic build, see tools/k2_phases.sh)."""
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
for nr, nc in ((56, 80), (56, 128), (56, 176), (28, 120), (8, 100)):
    rs = slots[:nr]; cs = np.concatenate([[0], slots[nr:nc - 2], rs, [1]]).astype(np.int32)
    before = np.zeros(4096, dtype=np.uint64)
    assert lib.fokl_debug_stamps_read(before.ctypes.data_as(ctypes.c_void_p), 4096) == 0
    for _ in range(20):
        ctx.gram(rs, cs, path=2)
    st = np.zeros(4096, dtype=np.uint64)
    assert lib.fokl_debug_stamps_read(st.ctypes.data_as(ctypes.c_void_p), 4096) == 0
    live = (st != before).reshape(512, 8).any(axis=1)      # the workgroups of THIS launch (the buffer keeps older entries)
    st = st.reshape(512, 2, 4).astype(float)
    chunks = (n + 31) // 32 / max(int(live.sum()), 1)
    for w, name in ((0, 'wavefront 0'), (1, 'wavefront 7')):
        per = np.median(st[live, w, :], axis=0) / chunks
        tot = per.sum()
        print(f'gram {nr} x {nc} {name}: {tot:7.0f} cycles per chunk ({int(live.sum())} workgroups, {chunks:.0f} chunks each): '
              f'issue {per[0]:6.0f} ({per[0] / tot:4.0%})  multiply {per[1]:6.0f} ({per[1] / tot:4.0%})  '
              f'wait for own pieces {per[2]:6.0f} ({per[2] / tot:4.0%})  barrier {per[3]:6.0f} ({per[3] / tot:4.0%})', flush=True)
