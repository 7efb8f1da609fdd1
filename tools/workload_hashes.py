"""sha256 of the synthetic datasets of bench.py's configurations: the golden fixtures of tests/golden/cfg*.npz were
computed in the build container from datasets regenerated from their seeds, so the GPU box must regenerate the very
same bits (numpy's Generator stream is platform independent; np.sin dispatches on CPU features)."""
import hashlib, os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import numpy as np
import bench
for cfg, unit, rows in ((2, 0, None), (4, 0, None), (4, 5, None), (1, 0, None), (3, 0, 100_000), (3, 0, None)):
    x, y, spec = bench.config_workload(cfg, unit, rows)
    print(cfg, unit, spec['rows'], hashlib.sha256(x.tobytes()).hexdigest()[:16], hashlib.sha256(y.tobytes()).hexdigest()[:16])
