"""A slice of tests/stress/random_parity.py inside the GPU suite: fifty random small problems (1-5 inputs, both kernels, 2- and
3-way, AIC / BIC, random chain lengths and tolerances) through the device path -- threaded native search, device chains, direct
kill decisions -- against the oracle's in-line search on the CPU: same model, BIC trace to 1e-9, draws to 1e-6, numpy's stream
on the same state.  (The 700-problem run is a hand-run stress job: profiles/random_parity_r05.txt.)"""
import importlib.util
import os

import pytest

from helpers import ROOT


@pytest.mark.gpu
def test_fifty_random_problems_match_the_oracle():
    spec = importlib.util.spec_from_file_location('random_parity', os.path.join(ROOT, 'tests', 'stress', 'random_parity.py'))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    bad = []
    for seed in range(50):
        ok, grown, t_ref, t_got, shape, kernel = rp.compare(seed, 3000)
        if not ok:
            bad.append((seed, shape, kernel, t_ref, t_got))
    assert bad == []
    assert rp.TALLY['kill_tests'] > 0
