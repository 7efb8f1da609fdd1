"""One warm fit of a bench configuration with the search's counters printed (development aid, GPU box).

    python tools/one_fit.py [config] [key ...]      keys: which fit_stats entries to print (default: all numeric ones)
"""
import os
import sys
import time
import warnings

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from fokl_gpy_amd import FoKLRoutines, getKernels


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    keys = sys.argv[2:]
    x, y, spec = bench.config_workload(config, 0, None)
    if spec['kernel'] == 'Cubic Splines':
        phis = getKernels.sp500()
    else:
        phis = getKernels.bernoulli()
        if spec['phis_cap']:
            phis = phis[:spec['phis_cap']]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=spec['kernel'], phis=phis, UserWarnings=False, ConsoleOutput=False, **spec['fit'])
        for rep in range(3):
            np.random.seed(spec['seed_fit'])
            t0 = time.perf_counter()
            model.fit(x, y, clean=True)
            dt = time.perf_counter() - t0
    st = model.fit_stats
    print(f"fit call {1e3 * dt:.1f} ms, search {1e3 * st['seconds']:.1f} ms, mtx {model.mtx.shape}")
    for k, v in st.items():
        if (not keys and isinstance(v, (int, float))) or k in keys:
            print(f"  {k:28s} {v}")


if __name__ == '__main__':
    main()
