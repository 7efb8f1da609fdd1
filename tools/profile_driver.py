"""cProfile of the Python driver thread over a few warm configs[2] searches (GPU box): where the interpreter's own time goes.

    python tools/profile_driver.py [fits]
"""
import cProfile
import io
import os
import pstats
import sys
import warnings

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from fokl_gpy_amd import FoKLRoutines, getKernels


def main():
    fits = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    x, y, spec = bench.config_workload(2, 0, None)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=spec['kernel'], phis=getKernels.bernoulli(), UserWarnings=False, ConsoleOutput=False,
                                  **spec['fit'])
        backend, n, m = model._prepare_fit(x, y, dict(clean=True))
        for _ in range(3):
            np.random.seed(spec['seed_fit'])
            model._search(backend, n, m)
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(fits):
            np.random.seed(spec['seed_fit'])
            model._search(backend, n, m)
        prof.disable()
    out = io.StringIO()
    st = pstats.Stats(prof, stream=out)
    st.sort_stats('tottime').print_stats(28)
    st.sort_stats('cumulative').print_stats(45)
    print(out.getvalue().replace(ROOT + '/', ''))
    print(f"per fit: {1e3 * model.fit_stats['seconds']:.1f} ms (under the profiler)")


if __name__ == '__main__':
    main()
