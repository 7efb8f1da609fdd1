"""SURVEY 8(f) N1 / N2 at the benchmark size (development aid): evaluate / coverage3 / bss_derivatives of the model fitted
on the configs[2] workload (N = 1e6, M = 8), wall time and the predict kernel's rate."""
import os, sys, time, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from fokl_gpy_amd import FoKLRoutines, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, y = bench.make_workload(12, n, 8)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False)
    np.random.seed(1000)
    t = time.time(); model.fit(x, y, clean=True); print('fit (incl. cleaning + upload) s', round(time.time() - t, 3))
    print('model', model.mtx.shape, 'draws', model.betas.shape)
    ctx = FoKLRoutines.device_backend(0).ctx
    for label, call in (('evaluate (mean)', lambda: model.evaluate()),
                        ('evaluate (mean + bounds)', lambda: model.evaluate(ReturnBounds=True)),
                        ('coverage3', lambda: model.coverage3()),
                        ('bss_derivatives d1 (8 inputs)', lambda: model.bss_derivatives()),
                        ('bss_derivatives d1 + d2', lambda: model.bss_derivatives(d1=True, d2=True))):
        call()                                                 # warm
        ctx.timing_enable(True); ctx.timing_reset()
        t = time.time(); out = call(); dt = time.time() - t
        tp, tb = ctx.timing_get(_capi.K_PREDICT), ctx.timing_get(_capi.K_BASIS)
        ctx.timing_enable(False)
        line = f'{label:32s} wall {dt * 1e3:8.1f} ms'
        if tp['launches']:
            line += (f" | predict kernel {tp['launches']} launch(es) {tp['ms']:7.2f} ms, {tp['flops'] / tp['ms'] / 1e9:6.1f} TFLOP/s, "
                     f"{tp['bytes'] / tp['ms'] / 1e6:7.1f} GB/s")
        if tb['launches']:
            line += f" | K1 {tb['launches']} launch(es) {tb['ms']:6.2f} ms, {tb['bytes'] / tb['ms'] / 1e6:7.1f} GB/s"
        print(line, flush=True)
