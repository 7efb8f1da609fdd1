"""
Golden vectors of the BASELINE configurations at (or near) their full size -- what bench.py times and what the
`-m gpu` tests of tests/test_config_goldens.py check.

    python tests/golden/make_config_golden.py cfg2 [cfg4 cfg1 cfg3 ...]

The real reference cannot produce these: its per-element Python loop (FoKLRoutines.py:1446-1485) needs ~8 us per
(row, term), i.e. about a day for the 1.0e10 basis evaluations of the configs[2] fit.  They come from the ORACLE
(oracle/fokl_oracle.py: the reference's algorithm statement for statement, pinned bit-exactly to the imported
reference by tests/golden/make_golden.py's fixtures at sizes the reference finishes), run here with

  * the REAL reference's ``clean`` (imported from /root/reference/src) for formatting + normalisation,
  * the sign-canonical ``eigh``,
  * the oracle's C column builder with the rows split over threads (element-wise identical to the scalar loop).

Stored per case: the workload's spec, sha256 of the raw and of the normalised dataset (the tests regenerate the
dataset from its seed), mtx, evs, the size / BIC of every gibbs() call in order, the kept draws, and numpy's global
stream after the fit.  BLAS thread count is recorded: at these N the last bits of XtX depend on it (tolerances in
the tests: mtx exact, evs 1e-9 relative, draws 1e-9 of the column's magnitude -- SURVEY 8(c)).
"""
import hashlib
import os
import sys
import time
import warnings

os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/src')

import numpy as np

from FoKL import FoKLRoutines as REF            # the reference itself: clean() only
from FoKL import getKernels as REF_GK

import bench                                    # the workloads are defined once, next to the benchmark
from fokl_gpy_amd import getKernels as own_gk
from oracle import fokl_oracle as O

THREADS = int(os.environ.get('GOLDEN_THREADS', '8'))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def run_case(name, config, unit=0, rows=None, **overrides):
    x, y, spec = bench.config_workload(config, unit, rows)
    fit_kw = dict(spec['fit'])
    fit_kw.update(overrides)
    kernel = spec['kernel']
    if kernel == 'Cubic Splines':
        tab = np.load(os.path.join(HERE, 'spline_phis.npz'))['table']
        phis, kid = own_gk.table_to_phis(tab), O.KERNEL_SPLINES
    else:
        phis, kid = tuple(REF_GK.bernoulli()), O.KERNEL_BERNOULLI
        ours = own_gk.bernoulli()
        assert all(np.array_equal(a, b) for a, b in zip(phis, ours))
        if spec['phis_cap']:
            phis = phis[:spec['phis_cap']]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = REF.FoKL(kernel=kernel, phis=phis, UserWarnings=False, ConsoleOutput=False)
        model.clean(x, y, _setattr=True)
        inputs, data = model.trainset()
    inputs, data = np.asarray(inputs, dtype=np.float64), np.asarray(data, dtype=np.float64)

    def build(xsm, phind, phis_, kernel_, terms):
        return O.build_columns_c(xsm, phind, phis_, kernel_, terms, threads=THREADS)

    trace = []
    np.random.seed(spec['seed_fit'])
    t0 = time.time()
    betas, mtx, evs = O.fit(inputs, data, phis, kid, eigh=O.eigh_canonical, build=build, trace=trace, **fit_kw)
    secs = time.time() - t0
    st = np.random.get_state()
    hp = dict(O.DEFAULT_HYPERS)
    hp.update(fit_kw)
    b, btau = O.default_b_btau(data, hp['a'], hp['atau'], hp['b'], hp['btau'])
    out = dict(config=config, unit=unit, rows=spec['rows'], inputs=spec['inputs'], kernel=kernel,
               phis_cap=spec['phis_cap'] or -1, seed=spec['seed'], seed_fit=spec['seed_fit'],
               fit_keys=np.array(list(fit_kw.keys()), dtype='U16'),
               fit_vals=np.array([float(v) for v in fit_kw.values()]),
               sha_raw_x=sha(x), sha_raw_y=sha(y), sha_norm_x=sha(inputs), sha_norm_y=sha(data),
               mtx=np.array(mtx, dtype=np.float64), evs=np.array(evs, dtype=np.float64), betas=np.array(betas),
               call_cols=np.array([t['cols'] for t in trace]), call_built=np.array([t['built'] for t in trace]),
               call_ev=np.array([t['ev'] for t in trace]), call_kill=np.array([t['kill'] for t in trace]),
               # what the kill tests read from each call's chain (round 5): the mean intercept draw over the second half
               # (FR:1671), and per sub-stage call the statistics of its new terms (FR:1656-1658), concatenated in call order
               call_b0=np.array([t['b0'] for t in trace]),
               stat_calls=np.array([i for i, t in enumerate(trace) if 'mean_abs' in t]),
               stat_sizes=np.array([t['mean_abs'].shape[0] for t in trace if 'mean_abs' in t]),
               stat_mean_abs=np.concatenate([t['mean_abs'] for t in trace if 'mean_abs' in t]),
               stat_rel_std=np.concatenate([t['rel_std'] for t in trace if 'rel_std' in t]),
               rng_key=st[1], rng_pos=st[2], rng_has_gauss=st[3], rng_cached=st[4], b=float(b), btau=float(btau),
               oracle_seconds=secs, oracle_threads=THREADS,
               blas_threads=os.environ.get('OPENBLAS_NUM_THREADS', 'default'))
    target = os.path.join(HERE, name + '.npz')
    if os.path.exists(target):
        # a regenerated fixture must say what the one in the repository says, bit for bit, in every field that one has (the
        # fields added since come on top); otherwise it is written next to it for a look
        old = np.load(target, allow_pickle=False)
        same = all(np.array_equal(np.asarray(old[k]), np.asarray(out[k])) for k in old.files
                   if k not in ('oracle_seconds', 'oracle_threads', 'blas_threads'))
        if not same:
            differing = [k for k in old.files if k not in ('oracle_seconds', 'oracle_threads', 'blas_threads')
                         and not np.array_equal(np.asarray(old[k]), np.asarray(out[k]))]
            target = os.path.join(HERE, name + '.regenerated.npz')
            print(f"[{name}] differs from the committed fixture in {differing}: written to {target}", flush=True)
    np.savez_compressed(target, **out)
    print(f"[{name}] {secs:.0f} s  terms={mtx.shape[0]} sub-stages={len(evs)} gibbs calls={len(trace)} "
          f"logical terms={int(np.sum(out['call_built']))}  max cols={int(np.max(out['call_cols']))}", flush=True)


CASES = {
    # the benchmarked workload itself: configs[2], uncapped, reference defaults
    'cfg2': lambda: run_case('cfg2_n1e6_m8', 2),
    # one unit of configs[4] (dataset seed 100, chain seed 1000)
    'cfg4': lambda: run_case('cfg4_unit0_n1e5_m8', 4, unit=0),
    'cfg4b': lambda: run_case('cfg4_unit5_n1e5_m8', 4, unit=5),
    # round 6: four more units of configs[4] (of its 64: 0, 5, 17, 29, 42, 63 are pinned) ...
    'cfg4c': lambda: run_case('cfg4_unit17_n1e5_m8', 4, unit=17),
    'cfg4d': lambda: run_case('cfg4_unit29_n1e5_m8', 4, unit=29),
    'cfg4e': lambda: run_case('cfg4_unit42_n1e5_m8', 4, unit=42),
    'cfg4f': lambda: run_case('cfg4_unit63_n1e5_m8', 4, unit=63),
    # ... and the configs[2] family on another dataset / chain seed (dataset seed 12 + 7, chain seed 1000 + 7), so that the
    # timed fit is not the only full-size Bernoulli search pinned
    'cfg2b': lambda: run_case('cfg2_unit7_n1e6_m8', 2, unit=7),
    # configs[1] at its full size and the reference's default draws
    'cfg1': lambda: run_case('cfg1_n1e5_m4_splines', 1),
    # configs[3] family (M = 16, 3-way, stages capped at 3) at the largest N / draws the oracle finishes in ~1/2 h:
    # its O(P^3) products per Gibbs iteration (FR:1521-1528) on 585-column models are what bounds it
    'cfg3': lambda: run_case('cfg3_n1e5_m16_way3', 3, rows=100_000, burnin=30, draws=30),
    # ... and at the benchmark's own N = 1e6 with the same shortened chains (the oracle's cost at this size is the 900
    # Gram matrices of up to 586 columns over a million rows and the column builds: 3.5 h here, most of it page faults
    # on the 4.7 GB design matrices the restatement copies per evaluation; with the configuration's 1000 + 1000 draws
    # the O(P^3) products per Gibbs iteration would add days)
    'cfg3big': lambda: run_case('cfg3_n1e6_m16_way3', 3, rows=1_000_000, burnin=30, draws=30),
    # ... and with chains long enough for the kill tests' Monte-Carlo statistics to mean something (round 4: 250 + 250 draws;
    # the O(P^3) products of 450 000 Gibbs iterations add about 2e14 flop to the 3.5 h above)
    'cfg3long': lambda: run_case('cfg3_n1e6_m16_way3_d250', 3, rows=1_000_000, burnin=250, draws=250),
}

if __name__ == '__main__':
    for key in sys.argv[1:] or ['cfg2']:
        CASES[key]()
