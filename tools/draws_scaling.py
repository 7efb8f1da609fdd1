"""How the fit time of the configs[2] workload responds to the length of the Gibbs chains (development aid): if the
serial random stream bounds the fit, halving burnin + draws should take close to half of the noise thread's time off."""
import os, sys, time, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from fokl_gpy_amd import FoKLRoutines
x, y = bench.make_workload(12, 1_000_000, 8)
for burnin, draws in ((1000, 1000), (500, 500), (250, 250), (2000, 2000)):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=1, burnin=burnin, draws=draws, UserWarnings=False, ConsoleOutput=False)
        be, n, m = model._prepare_fit(x, y, dict(clean=True))
        ts = []
        for rep in range(5):
            np.random.seed(1000)
            t = time.perf_counter(); model._search(be, n, m); ts.append(time.perf_counter() - t)
    st = model.fit_stats
    print(f"burnin+draws {burnin + draws}: best {1e3 * min(ts):.1f} ms, median {1e3 * sorted(ts)[2]:.1f} ms; evaluations "
          f"{st['gibbs_calls']}, noise busy {1e3 * st['pool_noise_s']:.1f} ms, queue wait {1e3 * st['noise_queue_wait_s']:.1f}, "
          f"verdict wait {1e3 * st['noise_verdict_wait_s']:.1f}; driver waits: eigh {1e3 * st['t_eigh']:.1f} chain "
          f"{1e3 * st['t_chain']:.1f} resid {1e3 * st['t_resid']:.1f}", flush=True)
