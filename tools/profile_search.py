"""Host-side profile of one forward-selection search on the configs[2] workload (development aid)."""
import cProfile, pstats, sys, os, time, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from fokl_gpy_amd import FoKLRoutines
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, y = bench.make_workload(12, n, 8)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False)
    be, n, m = model._prepare_fit(x, y, dict(clean=True))
    np.random.seed(1000); model._search(be, n, m)
    np.random.seed(1000)
    pr = cProfile.Profile(); pr.enable()
    t = time.time(); model._search(be, n, m); dt = time.time() - t
    pr.disable()
print('time', dt, model.fit_stats, 'final terms', model.mtx.shape, 'max order', model.mtx.max())
print('cols per call: max', max(t['cols'] for t in model.fit_trace), 'mean', np.mean([t['cols'] for t in model.fit_trace]))
print('evs', model.evs)
pstats.Stats(pr).sort_stats('tottime').print_stats(40)
