"""Where the first and last millisecond of a configs[2] search go (GPU box): pool creation, search teardown, pool teardown."""
import os, sys, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from fokl_gpy_amd import FoKLRoutines, getKernels

x, y, spec = bench.config_workload(2, 0, None)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = FoKLRoutines.FoKL(kernel=spec['kernel'], phis=getKernels.bernoulli(), UserWarnings=False, ConsoleOutput=False, **spec['fit'])
    backend, n, m = model._prepare_fit(x, y, dict(clean=True))
    acc = {}
    fits = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    for i in range(fits + 3):
        np.random.seed(spec['seed_fit'])
        model._search(backend, n, m)
        if i >= 3:
            for k, v in model.fit_stats.items():
                if k.startswith('t_') and isinstance(v, float):
                    acc[k] = acc.get(k, 0.0) + v
print({k: round(1e3 * v / fits, 3) for k, v in sorted(acc.items())})
