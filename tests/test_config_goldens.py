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
         'cfg3_n1e6_m16_way3', 'cfg3_n1e6_m16_way3_d250']


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
