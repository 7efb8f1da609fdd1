"""Fit time of the configs[2] family (M = 8, Bernoulli, 2-way, reference defaults) against the number of rows, one GPU
(DESIGN.md section 6 table): seconds per fit, device-kernel milliseconds, candidate terms per second, and the roofline
fractions of the three kernels.  FOKL_KILL_BIC / FOKL_K3 select the variants compared."""
import os, sys, time, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from fokl_gpy_amd import FoKLRoutines, _capi

rows = [int(float(a)) for a in sys.argv[1:]] or [100_000, 1_000_000, 10_000_000]
REPS = int(os.environ.get('N_SCALING_REPS', '5'))
TRACE = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'fokl_gram_trace_{os.getpid()}.txt')
os.environ['FOKL_GRAM_TRACE'] = TRACE      # one line per Gram launch: rows x columns and the roof it was booked under


flops_run_over_algorithmic = bench.mfma_flops_issued_over_algorithmic


for n in rows:
    x, y = bench.make_workload(12, n, 8)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False)
        be, nn, m = model._prepare_fit(x, y, dict(clean=True))
        del x, y
        for _ in range(int(os.environ.get('N_SCALING_WARMUP', '3'))):
            np.random.seed(1000); model._search(be, nn, m)
        ctx = be.ctx
        ctx.timing_enable(True); ctx.timing_reset()
        open(TRACE, 'w').close()
        ts = []
        for rep in range(REPS):
            np.random.seed(1000)
            t = time.perf_counter(); model._search(be, nn, m); ctx.sync(); ts.append(time.perf_counter() - t)
        ctx.timing_enable(False)
    st = model.fit_stats
    ks = {name: ctx.timing_get(kid) for name, kid in (('K1', _capi.K_BASIS), ('K3', _capi.K_RESID), ('K3mf', _capi.K_RESID_MF))}
    ks['K2'] = ctx.timing_get_gram()
    dev_ms = sum(k['ms'] for k in ks.values()) / REPS
    frac = {name: (k['ideal_ms'] / k['ms'] if k['ms'] > 0 else 0.0) for name, k in ks.items()}
    print(f"N={n:>11,d}: {min(ts):.3f} s/fit (median {sorted(ts)[len(ts) // 2]:.3f}), device kernels {dev_ms:.1f} ms/fit, "
          f"{st['terms_logical'] / min(ts):,.0f} candidate terms/s, evaluations {st['gibbs_calls']}, BIC from Gram "
          f"{st['bic_from_gram']}, matrix-free K3 {st['resid_matrix_free']}; roofline fractions "
          + ' '.join(f"{k} {v:.2f}" for k, v in frac.items())
          + f"; MFMA flops issued / algorithmic on the MFMA-bound Gram launches {flops_run_over_algorithmic(open(TRACE).read().splitlines()):.3f}"
          # (profiles/k2_clock_r05.txt: through launches of milliseconds the chip holds 1.92-1.98 GHz of the 2.4 the 78.6 TFLOP/s
          # peak is quoted at; through the 0.1-0.3 ms launches of N = 1e6 2.1-2.35)
          + (f"; K2 against the roof at the clock the chip holds (1.95 of 2.4 GHz): {frac['K2'] / (1.95 / 2.4):.2f}" if n >= 10_000_000 else ""),
          flush=True)
    ctx.upload(np.zeros((1, 1)), np.zeros(1), 1, np.zeros(2), 1, 2)          # let go of the big dataset
try:
    os.remove(TRACE)
except OSError:
    pass
