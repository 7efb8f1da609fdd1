import sys, os, warnings
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from fokl_gpy_amd import FoKLRoutines, getKernels
x, y, spec = bench.config_workload(2, 0, None)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = FoKLRoutines.FoKL(kernel=spec['kernel'], phis=getKernels.bernoulli(), UserWarnings=False, ConsoleOutput=False, **spec['fit'])
    backend, n, m = model._prepare_fit(x, y, dict(clean=True))
    for _ in range(4):
        np.random.seed(spec['seed_fit']); model._search(backend, n, m)
st = model.fit_stats
print({k: st[k] for k in st if 'forecast' in k or k in ('seconds', 'stats_by_chain_thread', 'chains_ahead', 'chains_ahead_unused')})
print('phases', {k: round(v*1e3, 2) for k, v in st['phases'].items()})
