"""K3 A/B on the device: stored-column residual pass vs the matrix-free pass (fokl_bic_resid_terms_launch) on
configs[2]-shaped models (N rows, M = 8, Bernoulli, 2-way): time per launch from HIP events, difference of the moments."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
from fokl_gpy_amd import _capi, getKernels, engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kid = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = 8
rng = np.random.default_rng(3)
x, y = rng.random((n, m)), rng.standard_normal(n)
ctx = _capi.DeviceContext(0)
if kid == 1:
    phis = getKernels.bernoulli()
else:
    phis = getKernels.table_to_phis(np.load(os.path.join(ROOT, 'tests', 'golden', 'spline_phis.npz'))['table'])
packed, nb, width = getKernels.pack_phis(phis, kid)
ctx.upload(x, y, kid, packed, nb, width)
shapes = {'9 cols (8 main, order 1)': [[1, 0]],
          '37 cols (orders 1 + (1,1))': [[1, 0], [1, 1]],
          '45 cols, U=16': [[1, 0], [1, 1], [2, 0]],
          '101 cols, U=16': [[1, 0], [1, 1], [2, 0], [2, 1]],
          '109 cols, U=24': [[1, 0], [1, 1], [2, 0], [2, 1], [3, 0]],
          '201 cols, U=40': [[1, 0], [1, 1], [2, 0], [2, 1], [3, 0], [2, 2], [3, 1], [4, 0], [5, 0]]}
ctx.timing_enable(True)
for label, pats in shapes.items():
    terms = np.vstack([engine.distinct_arrangements(p + [0] * (m - 2)) for p in pats]).astype(np.int32)
    T = terms.shape[0]
    ctx.reserve_slots(2 + T)
    slots = np.arange(2, 2 + T, dtype=np.int32)
    ctx.build_terms(terms, slots)
    beta = rng.standard_normal(T + 1)
    sl = np.concatenate([[0], slots]).astype(np.int32)
    want = ctx.bic_resid(sl, beta)
    try:
        ctx.bic_resid_terms_launch(terms, beta)
        got = ctx.bic_resid_fetch()
    except _capi.FoklNativeError:
        got = None                                            # outside the matrix-free pass's factor layouts
    ctx.sync(); ctx.timing_reset()
    for _ in range(20):
        ctx.bic_resid(sl, beta)
    for _ in range(20 if got is not None else 0):
        ctx.bic_resid_terms_launch(terms, beta); ctx.bic_resid_fetch()
    a, b = ctx.timing_get(_capi.K_RESID), ctx.timing_get(_capi.K_RESID_MF)
    line = f"{label:32s} columns {1e3 * a['ms'] / a['launches']:8.1f} us ({a['bytes'] / a['ms'] / 1e6:7.0f} GB/s)   "
    if got is None:
        line += "matrix-free: not offered (stored columns only)"
    else:
        line += (f"matrix-free {1e3 * b['ms'] / b['launches']:8.1f} us ({b['bytes'] / b['ms'] / 1e6:7.0f} GB/s of its own "
                 f"8 N (M_used + 1) bytes, {a['bytes'] / a['launches'] / (b['ms'] / b['launches']) / 1e6:7.0f} GB/s of the stored "
                 f"pass's 8 N (P + 2); {b['flops'] / b['ms'] / 1e9:6.2f} TFLOP/s)   relative difference of the moments: "
                 f"{abs(got[0] - want[0]) / max(abs(want[0]), 1e-300):.1e} {abs(got[1] - want[1]) / want[1]:.1e}")
    print(line, flush=True)
