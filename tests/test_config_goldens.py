"""
The BASELINE configurations at (or near) full size against oracle-generated goldens (tests/golden/cfg*.npz, made by
tests/golden/make_config_golden.py in the build container: the reference's own ``clean`` + the oracle's restatement
of FR:1350-1760 with the sign-canonical eigh).  The datasets are regenerated from their seeds (bench.config_workload)
and checked against the stored sha256 before anything is compared.

Tolerances: interaction matrix, the sequence of gibbs() calls (model size and number of columns the
reference builds for each -- the benchmark's numerator) and numpy's global stream after the fit: exact; BIC of every
call: 1e-9 relative; kept draws: 1e-9 of the column's largest magnitude.
"""
import hashlib
import os
import warnings

import numpy as np
import pytest

import bench
from helpers import GOLDEN, OracleBackend
from fokl_gpy_amd import FoKLRoutines, getKernels

CASES = ['cfg4_unit0_n1e5_m8', 'cfg4_unit5_n1e5_m8', 'cfg2_n1e6_m8', 'cfg1_n1e5_m4_splines', 'cfg3_n1e5_m16_way3',
         'cfg3_n1e6_m16_way3', 'cfg3_n1e6_m16_way3_d250',
         # round 6: four more of configs[4]'s 64 units, and the configs[2] family on another dataset / chain seed
         'cfg4_unit17_n1e5_m8', 'cfg4_unit29_n1e5_m8', 'cfg4_unit42_n1e5_m8', 'cfg4_unit63_n1e5_m8', 'cfg2_unit7_n1e6_m8']


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def load_golden(name):
    path = os.path.join(GOLDEN, name + '.npz')
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz has not been generated (tests/golden/make_config_golden.py)")
    return np.load(path, allow_pickle=False)


def fit_like_golden(g, backend=None):
    """The product's fit on the golden's workload.  -> (model, betas, mtx, evs, numpy state after the fit)"""
    x, y, spec = bench.config_workload(int(g['config']), int(g['unit']), int(g['rows']))
    assert _sha(x) == str(g['sha_raw_x']) and _sha(y) == str(g['sha_raw_y']), "dataset does not regenerate bit for bit"
    fit_kw = {str(k): (bool(v) if str(k) in ('way3', 'gimmie', 'aic') else
                       int(v) if str(k) in ('burnin', 'draws', 'tolerance') else float(v))
              for k, v in zip(g['fit_keys'], g['fit_vals'])}
    init = dict(kernel=str(g['kernel']), UserWarnings=False, ConsoleOutput=False)
    if str(g['kernel']) == 'Cubic Splines':
        init['phis'] = getKernels.table_to_phis(np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table'])
    elif int(g['phis_cap']) > 0:
        init['phis'] = getKernels.bernoulli()[:int(g['phis_cap'])]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(**init)
        if backend is not None:
            model._backend_override = backend
        np.random.seed(int(g['seed_fit']))
        betas, mtx, evs = model.fit(x, y, clean=True, **fit_kw)
    assert _sha(model.inputs) == str(g['sha_norm_x']) and _sha(model.data) == str(g['sha_norm_y']), \
        "clean() normalises differently from the reference's"
    return model, betas, mtx, evs, np.random.get_state()


def assert_matches_golden(g, model, betas, mtx, evs, state, draws_tol=1e-9):
    assert mtx.shape == g['mtx'].shape and np.array_equal(mtx, g['mtx']), "selected interaction matrix differs"
    trace = model.fit_trace
    assert [t['cols'] for t in trace] == g['call_cols'].tolist(), "sequence of gibbs() calls differs"
    assert [t['built'] for t in trace] == g['call_built'].tolist()
    assert [t['kill'] for t in trace] == g['call_kill'].astype(bool).tolist()
    assert model.fit_stats['terms_logical'] == int(np.sum(g['call_built']))
    np.testing.assert_allclose([t['ev'] for t in trace], g['call_ev'], rtol=1e-9)
    np.testing.assert_allclose(evs, g['evs'], rtol=1e-9)
    assert betas.shape == g['betas'].shape
    scale = np.max(np.abs(g['betas']), axis=0)
    assert np.max(np.abs(betas - g['betas']) / scale) < draws_tol
    assert np.array_equal(state[1], g['rng_key']) and state[2] == int(g['rng_pos'])
    assert state[3] == int(g['rng_has_gauss']) and state[4] == float(g['rng_cached'])
    assert abs(model.b - float(g['b'])) <= 1e-15 * abs(float(g['b']))
    assert abs(model.btau - float(g['btau'])) <= 1e-15 * abs(float(g['btau']))
    assert_chain_statistics_match(g, model)


def assert_chain_statistics_match(g, model, tol=1e-9):
    """What the kill tests READ from the chains, not only what they decided (fixtures regenerated in round 5 carry it):
    per sub-stage the |mean beta| and std / |mean| of its new terms (FR:1656-1658) against the oracle's, 1e-9 of the
    largest |mean beta| of the sub-stage / 1e-9 relative where the ratio is not ill-conditioned; per gibbs() call whose
    chain the search looked at, the mean intercept draw of FR:1671 -- for a kill test that is the chain of a model whose
    eigenpairs were derived from the model it was tested against and which ran on the device.  -> how many intercept
    means were compared."""
    if 'call_b0' not in g.files:
        return 0
    stats = model.fit_substage_stats
    assert len(stats) == g['stat_sizes'].shape[0]
    at = 0
    for st, size in zip(stats, g['stat_sizes']):
        want_mean, want_rel = g['stat_mean_abs'][at:at + size], g['stat_rel_std'][at:at + size]
        at += int(size)
        scale = max(float(np.max(want_mean)), 1e-300)
        assert st['mean_abs'].shape == want_mean.shape
        assert np.max(np.abs(st['mean_abs'] - want_mean)) <= tol * scale
        # std / |mean|: a mean within 1e-3 of its sub-stage's largest may be cancellation noise relative to itself
        firm = want_mean > 1e-3 * scale
        assert np.all(np.abs(st['rel_std'][firm] - want_rel[firm]) <= 1e-6 * np.abs(want_rel[firm]))
    have = np.array([t.get('b0', np.nan) for t in model.fit_trace])
    seen = ~np.isnan(have)
    want = g['call_b0']
    assert have.shape == want.shape
    assert np.all(np.abs(have[seen] - want[seen]) <= tol * np.abs(want[seen]))
    # (which of the accepted kill tests' chains the search looked at -- guessed decisions are confirmed against them, the
    # others never run -- depends on thread timing: the count is returned, not asserted)
    return int(np.count_nonzero(seen))


def test_config_datasets_regenerate_bit_for_bit():
    seen = 0
    for name in CASES:
        path = os.path.join(GOLDEN, name + '.npz')
        if not os.path.exists(path):
            continue
        g = np.load(path)
        x, y, _ = bench.config_workload(int(g['config']), int(g['unit']), int(g['rows']))
        assert _sha(x) == str(g['sha_raw_x']) and _sha(y) == str(g['sha_raw_y']), name
        seen += 1
    assert seen >= 1


@pytest.mark.parametrize('name', ['cfg4_unit0_n1e5_m8', 'cfg1_n1e5_m4_splines'])
def test_host_logic_reproduces_full_size_configs_on_the_checker_backend(name):
    """The search driver + native sampler on the CPU stand-in backend (tests/helpers.OracleBackend) against the
    goldens of one configs[4] unit (N = 1e5, M = 8, Bernoulli) and of configs[1] (N = 1e5, M = 4, Cubic Splines) at
    their full size and 2000 Gibbs iterations per evaluation."""
    g = load_golden(name)
    assert_matches_golden(g, *fit_like_golden(g, OracleBackend()))


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_full_size_config_on_gpu_matches_the_golden(name):
    """configs[2] (the benchmarked fit itself, uncapped, reference defaults), two configs[4] units, configs[1] at
    1000 + 1000 draws and the configs[3] family through the HIP path."""
    g = load_golden(name)
    assert_matches_golden(g, *fit_like_golden(g))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['cfg2_n1e6_m8', 'cfg1_n1e5_m4_splines', 'cfg4_unit0_n1e5_m8'])
def test_full_size_config_with_the_eigen_decompositions_on_the_device(monkeypatch, name):
    """FOKL_EIGH=device: G2 of every model of up to 192 columns by the Jacobi kernels (csrc/fokl_spectral_device.inc)
    instead of LAPACK on host threads.  Its eigenvectors differ from dsyevr's within their conditioning (both from the
    exact ones: tests/test_spectral_device.py), which reaches the draws at the 1e-10 level of the column scale -- the
    selected model, every gibbs() call, every kill-test decision and numpy's stream are the golden's, the draws within
    the same 1e-9."""
    g = load_golden(name)
    monkeypatch.setenv('FOKL_EIGH', 'device')
    model, betas, mtx, evs, state = fit_like_golden(g)
    assert model.fit_stats['eigh_mode'] == 'device' and model.fit_stats['spectral_device'] > 0
    assert_matches_golden(g, model, betas, mtx, evs, state)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['cfg2_n1e6_m8', 'cfg4_unit5_n1e5_m8'])
def test_kill_tests_eigenpairs_derived_from_the_tested_against_model(monkeypatch, name):
    """The default: G2 of a kill test's model follows from the eigenpairs of the model it is tested against (secular equation
    + one product, fokl_pool_submit_spectral_update) -- most of a fit's G2 jobs -- and the fit is the golden's: model, every
    gibbs() call, numpy's stream exactly, draws within the same 1e-9.  FOKL_EIGH_UPDATE=0 (every model decomposed afresh)
    gives the same fit; the two differ in the draws' last digits only."""
    g = load_golden(name)
    model, betas, mtx, evs, state = fit_like_golden(g)
    st = model.fit_stats
    assert st['eigh_update_from'] == 8 and st['spectral_updated'] > 0.5 * st['spectral_submitted']
    assert_matches_golden(g, model, betas, mtx, evs, state)
    monkeypatch.setenv('FOKL_EIGH_UPDATE', '0')
    model0, betas0, mtx0, evs0, state0 = fit_like_golden(g)
    assert 'eigh_update_from' not in model0.fit_stats and model0.fit_stats['spectral_updated'] == 0
    assert_matches_golden(g, model0, betas0, mtx0, evs0, state0)
    assert np.array_equal(mtx, mtx0) and betas.shape == betas0.shape
    assert np.abs(betas - betas0).max() <= 1e-9 * np.abs(betas0).max()


# The knobs that change which code runs a fit (tools/knob_suite.sh walks all of them over every fit-level test by hand;
# the default-changing ones are part of the GPU suite here, each on one golden): every mode must give the golden's model,
# calls, stream, BICs and draws.
MODES = [
    (('FOKL_KILL_DECIDE', 'g2'),),                  # kill tests decided from G2 of every trial model (round 4's loop)
    (('FOKL_EIGH_UPDATE', '0'),),                   # every model decomposed afresh
    (('FOKL_CHAIN', 'host'),),                      # kill tests' chains on host threads
    (('FOKL_SEARCH', 'python'),),                   # engine.py's statement of the kill-test loop
    (('FOKL_EIGH', 'device'),),                     # G2 by the Jacobi kernels
    (('FOKL_G2_DEFER_FROM', '8'),),                 # G2 of accepted models only when something needs it
    (('FOKL_SPECULATE_ACROSS', '0'), ('FOKL_SPECULATION', '4')),   # a short order book, nothing ordered across the boundary
    (('FOKL_DCHAIN_ROWS', '0'),),                   # tapes materialised on the host, read over the bus
]


@pytest.mark.gpu
@pytest.mark.parametrize('mode', MODES, ids=lambda m: '+'.join(f'{k}={v}' for k, v in m))
def test_default_changing_modes_give_the_golden_fit(monkeypatch, mode):
    g = load_golden('cfg4_unit0_n1e5_m8')
    for key, value in mode:
        monkeypatch.setenv(key, value)
    model, betas, mtx, evs, state = fit_like_golden(g)
    st = model.fit_stats
    if ('FOKL_SEARCH', 'python') in mode:
        assert st['search_driver'] == 'python'
    else:
        assert st['search_driver'] == 'native'
        assert st['kill_decide'] == ('g2' if ('FOKL_KILL_DECIDE', 'g2') in mode else 'direct')
    if ('FOKL_CHAIN', 'host') in mode:
        assert st['chain_mode'] == 'host'
    if ('FOKL_EIGH_UPDATE', '0') in mode:
        assert st['spectral_updated'] == 0
    if ('FOKL_EIGH', 'device') in mode:
        assert st['eigh_mode'] == 'device' and st['spectral_device'] > 0
    assert_matches_golden(g, model, betas, mtx, evs, state)


@pytest.mark.gpu
def test_direct_kill_decisions_are_confirmed_by_the_eigenpairs():
    """The default: a kill test's BIC comes from the sub-stage's least-squares model downdated column by column; every
    accepted model's eigenpairs bring a second BIC (Gram identity on their betahat) that is held against it.  On the
    benchmarked fit: every kill test decided that way, the two BICs within 1e-12, a third of the accepted models replaced
    before anything looked at their draws (no chain), and the chains that did run confirm the guessed decisions."""
    g = load_golden('cfg2_n1e6_m8')
    model, betas, mtx, evs, state = fit_like_golden(g)
    st = model.fit_stats
    assert st['kill_decide'] == 'direct' and st['direct_tests'] == st['kill_tests'] > 300
    assert st['direct_max_rel'] < 1e-12
    # (how many accepted models were replaced before anything looked at their draws -- st['chains_cancelled'] -- hangs on
    # whether a model's G2 arrives before or after its replacement: thread timing, reported by bench.py, not asserted)
    assert st['guesses_verified'] == st['guessed'] > 100
    assert st['searches_repeated'] == 0
    assert_matches_golden(g, model, betas, mtx, evs, state)


@pytest.mark.gpu
def test_fit_with_a_chain_engine_of_few_slots_falls_back_to_host_chains(monkeypatch):
    """Every device slot alive (here: an engine of six): a kill test's chain then runs on a host thread from the tape
    materialised after all -- including chains of models whose second clause had been GUESSED as for a device chain (they
    were decided before their chain existed): their statistics must still confirm the guesses before the outcome goes
    (round 5: such an outcome used to be released with its checks open, and the final verification failed)."""
    from fokl_gpy_amd import host_pipeline
    g = load_golden('cfg4_unit0_n1e5_m8')
    host_pipeline.close_chain_engines()
    monkeypatch.setenv('FOKL_DCHAIN_SLOTS', '6')
    try:
        model, betas, mtx, evs, state = fit_like_golden(g)
        st = model.fit_stats
        assert st['tapes_materialised'] > 20 and st['guesses_verified'] == st['guessed'] > 0
        assert st['searches_repeated'] == 0
        assert_matches_golden(g, model, betas, mtx, evs, state)
    finally:
        host_pipeline.close_chain_engines()
