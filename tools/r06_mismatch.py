"""Round 6: what differs on the random-parity problems that are reported as mismatches although the models agree."""
import os, sys, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'stress')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import random_parity as rp
warnings.simplefilter('ignore')
for seed in [int(a) for a in sys.argv[1:]]:
    x, y, kw = rp.problem(seed)
    ref = rp.fit(x, y, kw, seed, True)
    got = rp.fit(x, y, kw, seed, False)
    same_model = ref[1].shape == got[1].shape and np.array_equal(ref[1], got[1])
    scale = np.abs(ref[0]).max()
    draws = np.max(np.abs(got[0] - ref[0])) / scale if ref[0].shape == got[0].shape else None
    evs = np.max(np.abs(np.asarray(got[2]) - np.asarray(ref[2])) / np.abs(np.asarray(ref[2]))) if len(ref[2]) == len(got[2]) else None
    print(seed, x.shape, {k: v for k, v in kw.items() if k != 'phis'}, 'same model', same_model, 'stream equal', ref[3] == got[3],
          'evs rel', evs, 'draws / scale', draws, 'evs', len(ref[2]), len(got[2]), flush=True)
