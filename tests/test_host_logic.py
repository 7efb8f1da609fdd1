"""Host side of the product on CPU: search driver, Gram cache, slot bookkeeping, sampler hand-off, class surface.

The device backend is replaced by ``helpers.OracleBackend`` (tests only -- the product has no CPU path), so what
is checked here is everything *around* the HIP kernels, against the reference fixtures:
selected interaction matrix exactly, BIC trace <= 1e-9 relative, posterior draws <= 1e-9 * max|column|
(1e-6 on the degenerate sigmoid grid), identical consumption of numpy's global random stream.
"""
import os
import pickle
import warnings

import numpy as np
import pytest

from helpers import GOLDEN, FIT_CASES, OracleBackend, load_case, UNITS_PATH
from fokl_gpy_amd import FoKLRoutines, engine, host_pipeline, _capi, getKernels
from oracle import fokl_oracle as O


def make_model(kname, phis, hy):
    model = FoKLRoutines.FoKL(kernel=kname, phis=phis, UserWarnings=False, ConsoleOutput=False, **hy)
    model._backend_override = OracleBackend()
    return model


def fit_case(name):
    g, hy, kname, kid, phis = load_case(name)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = make_model(kname, phis, hy)
        np.random.seed(int(g['seed']))
        betas, mtx, evs = model.fit(g['raw_inputs'], g['raw_data'], clean=True)
    return g, model, betas, mtx, evs


def rng_fingerprint():
    import hashlib
    st = np.random.get_state()
    return float(int(hashlib.sha256(st[1].tobytes()).hexdigest()[:16], 16) % (2 ** 53)), st[2], st[3], st[4]


REGULAR = [c for c in FIT_CASES if not c.startswith('testdata10')]


@pytest.mark.parametrize('name', REGULAR)
def test_fit_matches_reference(name):
    if not os.path.exists(os.path.join(GOLDEN, name + '.npz')):
        pytest.skip('fixture not generated')
    g, model, betas, mtx, evs = fit_case(name)
    assert mtx.dtype == np.float64 and mtx.shape == g['canon_mtx'].shape
    assert np.array_equal(mtx, g['canon_mtx'])                                   # bit-exact selection
    assert len(evs) == len(g['canon_evs'])
    assert np.max(np.abs(evs - g['canon_evs']) / np.abs(g['canon_evs'])) < 1e-9  # BIC trace
    gb = g['canon_betas']
    tol = 1e-6 if name == 'sigmoid_splines' else 1e-9
    assert betas.shape == gb.shape
    assert np.max(np.abs(betas - gb) / np.max(np.abs(gb), axis=0)) < tol         # posterior draws
    # the global numpy stream ends exactly where the reference's does
    fp = rng_fingerprint()
    assert fp[0] == g['canon_rng_after_fit'][0] and fp[1] == g['canon_rng_after_fit'][1]
    assert fp[2] == g['canon_rng_after_fit'][2] and fp[3] == float(g['canon_rng_cache_after_fit'])
    # same number and sizes of model evaluations as the reference made
    assert [t['cols'] for t in model.fit_trace] == g['canon_gibbs_sizes'].tolist()
    assert model.fit_stats['terms_logical'] == sum(t['built'] for t in model.fit_trace)
    # side effects of fit (FR:1316-1317, 1344, 1348, 1755-1758)
    assert np.array_equal(model.inputs, g['canon_norm_inputs']) and np.array_equal(model.data, g['canon_norm_data'])
    assert np.allclose(np.array(model.minmax, dtype=float), g['canon_minmax'])
    assert model.b == pytest.approx(float(g['canon_b']), rel=1e-14)
    assert model.btau == pytest.approx(float(g['canon_btau']), rel=1e-14)
    assert np.allclose(model.avg_betas, np.mean(betas, axis=0))


@pytest.mark.parametrize('name', [c for c in REGULAR if c not in ('bern_m8_capped', 'splines_m4')])
def test_coverage3_matches_reference(name):
    if not os.path.exists(os.path.join(GOLDEN, name + '.npz')):
        pytest.skip('fixture not generated')
    g, model, betas, mtx, evs = fit_case(name)
    if 'canon_cov_mean' not in g.files:
        pytest.skip('no coverage fixture')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        mean, bounds, rmse = model.coverage3()
    assert np.array_equal(model.setnos, g['canon_setnos'])           # np.random.choice consumed identically
    scale = np.max(np.abs(g['canon_cov_mean']))
    tol = 1e-6 if name == 'sigmoid_splines' else 1e-10
    assert np.max(np.abs(mean - g['canon_cov_mean'])) < tol * scale
    assert np.max(np.abs(bounds - g['canon_cov_bounds'])) < tol * scale
    assert mean.shape == (model.inputs.shape[0],) and bounds.shape == (model.inputs.shape[0], 2)
    assert np.shape(rmse) == ()
    assert abs(rmse - float(g['canon_cov_rmse'])) < 1e-9


@pytest.mark.parametrize('name', ['testdata10_default', 'testdata10_changed'])
def test_reference_test_dataset_until_saturation(name):
    """The reference's own 10-row test set (test/testdatatest.csv, seeds of test/makingdata.py).  With 10 rows the
    model saturates (P + 1 >= N) after 7 sub-stages and XtX becomes numerically singular; from there the reference's
    numbers are rounding noise of its own BLAS.  Parity is pinned on everything before that point."""
    g, model, betas, mtx, evs = fit_case(name)
    ge = g['canon_evs']
    k = 7
    assert np.max(np.abs(evs[:k] - ge[:k]) / np.abs(ge[:k])) < 1e-9
    sizes = g['canon_gibbs_sizes'].tolist()
    upto = sizes.index(10) if 10 in sizes else len(sizes)
    assert [t['cols'] for t in model.fit_trace][:upto] == sizes[:upto]
    assert betas.shape[0] == 1000 and mtx.shape[1] == 2


# ---------------------------------------------------------------------------------------------------------
# driver pieces
# ---------------------------------------------------------------------------------------------------------

def test_enumeration_matches_reference_vectors():
    units = np.load(UNITS_PATH)
    for key in units.files:
        if key.startswith('enum_'):
            pattern = [int(v) for v in key[len('enum_'):].split('_')]
            assert np.array_equal(engine.distinct_arrangements(pattern), units[key]), key


def test_indvec_progression_matches_oracle():
    for m, way3 in ((1, False), (2, False), (5, False), (4, True), (8, True)):
        sett = 1 if m == 1 else (3 if way3 else 2)
        for ind in range(1, 9):
            assert np.array_equal(engine.deal_indvec(ind, m, sett), O.deal_indvec(ind, m, sett))
    seq = []
    v = engine.deal_indvec(6, 5, 3)
    while True:
        seq.append(v.copy().tolist())
        if not engine.advance_indvec(v, 5, True):
            break
    assert seq == [[2, 2, 2, 0, 0], [3, 2, 1, 0, 0], [4, 1, 1, 0, 0], [4, 2, 0, 0, 0], [5, 1, 0, 0, 0],
                   [6, 0, 0, 0, 0]]


def test_way3_with_two_inputs_raises_like_the_reference():
    with pytest.raises(IndexError):
        v = engine.deal_indvec(1, 2, 3)
        engine.advance_indvec(v, 2, True)


def test_slot_pool_never_hands_out_reserved_slots():
    be = OracleBackend()
    be.upload(np.zeros((4, 1)), np.zeros(4), O.KERNEL_BERNOULLI, *getKernels.pack_phis(getKernels.bernoulli(), 1))
    pool = engine.SlotPool(be, initial=8)
    got = pool.take(6)
    assert sorted(got) == [2, 3, 4, 5, 6, 7]
    more = pool.take(5)                       # forces growth
    assert min(more) >= 8 and be.capacity >= 13
    pool.give(got[:2])
    assert sorted(pool.take(2)) == sorted(got[:2])


def test_kill_tests_reuse_the_substage_gram(monkeypatch):
    """One basis build and one Gram block per sub-stage; every other model evaluation is a sub-matrix lookup.
    The pipelined search builds the coming sub-stage early, so a search ended by the stop rule has built one
    sub-stage it did not use (FOKL_FORESIGHT=0 switches that off)."""
    monkeypatch.setenv('FOKL_NOISE_PIPELINE', '1')                 # these are features of the threaded search
    for foresight, spare in (('0', 0), ('8', 1)):
        monkeypatch.setenv('FOKL_FORESIGHT', foresight)
        g, model, *_ = fit_case('bern_m3')
        be = model._backend_override
        st = model.fit_stats
        assert be.calls['build'] == st['substages'] + spare
        assert be.calls['gram'] == st['substages'] + 1 + spare   # + the [ones, y] seed block
        assert st['gibbs_calls'] == st['substages'] + st['kill_tests']
        assert st['terms_physical'] < st['terms_logical']
        if spare:
            assert st['forecasts_used'] > 0


def test_building_the_coming_substage_early_changes_nothing(monkeypatch):
    """FOKL_FORESIGHT: K1 / K2 of the next sub-stage and G2 of its predicted model run before the current kill tests
    are over.  Same search, same random stream; the Gram entries may differ in the last bits (a BLAS / K2 call's
    summation order depends on which other columns share the call)."""
    monkeypatch.setenv('FOKL_NOISE_PIPELINE', '1')                 # these are features of the threaded search
    runs = []
    for foresight in ('0', '8', '1000'):
        monkeypatch.setenv('FOKL_FORESIGHT', foresight)
        g, model, betas, mtx, evs = fit_case('bern_m6')
        runs.append((betas, mtx, evs, rng_fingerprint(), model.fit_stats['forecasts_used']))
    assert runs[0][4] == 0 and runs[1][4] > 0
    for other in runs[1:]:
        assert np.array_equal(other[1], runs[0][1]) and other[3] == runs[0][3]
        np.testing.assert_allclose(other[2], runs[0][2], rtol=1e-12)
        np.testing.assert_allclose(other[0], runs[0][0], rtol=1e-8, atol=1e-10)


# ---------------------------------------------------------------------------------------------------------
# class surface (reference: FoKLRoutines.py constructor / fit / evaluate keyword handling, test/test_FoKL.py)
# ---------------------------------------------------------------------------------------------------------

def test_constructor_defaults_and_unknown_keyword():
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False)
    assert model.kernel == 'Bernoulli Polynomials' and len(model.phis) == 20 and len(model.phis[3]) == 5
    assert (model.a, model.atau, model.tolerance, model.burnin, model.draws) == (4, 4, 3, 1000, 1000)
    assert model.b is None and model.btau is None and model.relats_in == [] and model.setnos is None
    assert (model.threshav, model.threshstda, model.threshstdb) == (0.05, 0.5, 2)
    with pytest.raises(ValueError, match="Unexpected keyword argument: 'backend'"):
        FoKLRoutines.FoKL(backend='hip')
    assert FoKLRoutines.FoKL(kernel=1, aic='on', way3='off', UserWarnings='no').aic is True


def test_default_kernel_is_cubic_splines_with_500_bases():
    model = FoKLRoutines.FoKL(UserWarnings=False)
    assert model.kernel == 'Cubic Splines' and len(model.phis) == 500
    assert len(model.phis[0]) == 4 and model.phis[0][0].shape == (499,)


def test_fit_rejects_unknown_keyword_and_needs_data():
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False)
    model._backend_override = OracleBackend()
    with pytest.raises(ValueError, match="Unexpected keyword argument"):
        model.fit(np.zeros((5, 1)), np.zeros(5), clean=True, device=0)
    with pytest.raises(IndexError):        # the reference formats `None` before reporting (FR:1275 -> FR:292)
        model.fit(clean=True)


def test_fit_accepts_pandas_and_returns_ndarrays():
    """test/test_FoKL.py::test_fit_for_shaping: outputs are ndarrays even for pandas inputs."""
    import pandas as pd
    rng = np.random.default_rng(3)
    df = pd.DataFrame({'x': rng.random(80), 'y': rng.random(80)})
    data = pd.Series(np.sin(3 * df['x']) + 0.1 * rng.standard_normal(80))
    model = FoKLRoutines.FoKL(kernel=1, burnin=40, draws=40, UserWarnings=False, ConsoleOutput=False)
    model._backend_override = OracleBackend()
    np.random.seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        betas, mtx, evs = model.fit(df, data, clean=True)
    assert isinstance(betas, np.ndarray) and isinstance(mtx, np.ndarray) and isinstance(evs, np.ndarray)
    assert betas.shape == (40, mtx.shape[0] + 1) and mtx.shape[1] == 2
    assert model.inputs.shape == (80, 2) and model.data.shape == (80, 1)
    assert model.inputs.min() == 0.0 and model.inputs.max() == 1.0


def test_hyperparameters_can_be_overridden_in_fit():
    rng = np.random.default_rng(4)
    x = rng.random((60, 2))
    y = x[:, 0] + 0.05 * rng.standard_normal(60)
    model = FoKLRoutines.FoKL(kernel=1, burnin=30, draws=30, UserWarnings=False, ConsoleOutput=False)
    model._backend_override = OracleBackend()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model.fit(x, y, clean=True, aic='on', a=3, b=1.8, atau=17, btau=2100.5, tolerance=1)
    assert model.aic is True and (model.a, model.b, model.atau, model.btau, model.tolerance) == (3, 1.8, 17, 2100.5, 1)


def test_relats_in_behaviour_matches_reference_quirks():
    x = np.random.default_rng(0).random((30, 3))
    y = x[:, 0]
    for relats, exc in (([[1, 0, 0]], TypeError), ([1, 0, 1], NameError)):
        model = FoKLRoutines.FoKL(kernel=1, relats_in=relats, UserWarnings=False, ConsoleOutput=False)
        model._backend_override = OracleBackend()
        with pytest.raises(exc):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                model.fit(x, y, clean=True)


def test_out_of_scope_entry_points_say_so():
    """to_pyomo (out of scope) refuses loudly; sequential updating is built (tests/test_fitupdate.py)."""
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False)
    with pytest.raises(NotImplementedError):
        model.to_pyomo()


def test_evaluate_requires_minmax_and_validates_draws():
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False)
    with pytest.raises(ValueError, match='minmax'):
        model.evaluate(np.zeros((3, 1)))
    g, model, betas, mtx, evs = fit_case('bern_m1')
    with pytest.raises(ValueError, match='exceeds the number of draws'):
        model.evaluate(model.inputs, draws=10 ** 6)


def test_evaluate_basis_api_matches_reference_values():
    units = np.load(UNITS_PATH)
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False)
    c = model.phis[4]
    vals = [model.evaluate_basis(c, np.float64(x)) for x in units['bern_x'][:20]]
    assert np.array_equal(np.array(vals), units['bern_vals'][4, :20])
    assert model.evaluate_basis([1.0, 2.0, 3.0, 4.0], 2.0, kernel=0, d=1) == 2 + 2 * 3 * 2 + 3 * 4 * 4
    assert model.evaluate_basis([1.0, 2.0, 3.0, 4.0], 2.0, kernel='Cubic Splines', d=2) == 2 * 3 + 6 * 4 * 2
    with pytest.raises(ValueError):
        model.evaluate_basis(c, 0.5, kernel='Fourier')


def test_clean_formats_like_the_reference():
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        x, y = model.clean([[1.0, 2.0, 3.0, 4.0], [10.0, 20.0, 30.0, 50.0]], [1, 2, 3, 4])
    assert x.shape == (4, 2) and y.shape == (4, 1) and y.dtype == np.float64        # auto-transposed
    assert model.minmax == [[1.0, 4.0], [10.0, 50.0]] and x[:, 1].tolist() == [0.0, 0.25, 0.5, 1.0]
    assert model.trainlog is None
    with pytest.raises(ValueError, match="'data' must be a vector"):
        model._format(np.zeros((3, 2)), np.zeros((3, 2)))
    with pytest.raises(ValueError):
        model.clean(np.zeros((4, 1)), pillow=[0.1, 0.1])    # the reference rejects the 2-element flat form too
    m2 = FoKLRoutines.FoKL(kernel=1, UserWarnings=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        xp = m2.clean(np.array([[0.0], [10.0]]), pillow=0.1)
    assert np.allclose(m2.minmax, [[-1.0, 11.0]]) and np.allclose(xp[:, 0], [1 / 12, 11 / 12])


def test_trainlog_fraction_uses_the_global_stream():
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False)
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model.clean(np.arange(50.0)[:, None], np.arange(50.0), train=0.5)
    assert model.trainlog.dtype == bool and model.trainlog.sum() == 25
    xi, yi = model.trainset()
    assert xi.shape == (25, 1) and yi.shape == (25, 1)


def test_save_load_roundtrip_keeps_results(tmp_path):
    g, model, betas, mtx, evs = fit_case('bern_m1')
    path = model.save('m', str(tmp_path))
    assert path.endswith('m.fokl')
    again = FoKLRoutines.load(path)
    assert np.array_equal(again.betas, betas) and np.array_equal(again.mtx, mtx) and again.kernel == model.kernel
    assert not hasattr(again, '_backend_override')
    pickle.dumps(model.fit_stats)


def test_clear_keeps_hyperparameters():
    g, model, *_ = fit_case('bern_m1')
    model.clear()
    assert not hasattr(model, 'betas') and hasattr(model, 'phis') and hasattr(model, 'a')


def test_product_has_no_cpu_fallback():
    """Without the override hook the class goes to the HIP backend, which must raise on a GPU-less host."""
    if _capi.device_count() > 0:
        pytest.skip('a GPU is present')
    FoKLRoutines._CONTEXTS.clear()
    model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False)
    with pytest.raises(_capi.FoklNativeError):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model.fit(np.random.default_rng(0).random((20, 1)), np.zeros(20), clean=True)


def test_kill_test_bic_from_gram_matches_explicit_residuals(monkeypatch):
    """FOKL_KILL_BIC=check: every kill-test candidate's BIC both ways (Gram identity in extended precision vs the
    backend's residual pass); the search itself is unchanged by the choice."""
    monkeypatch.setenv('FOKL_NOISE_PIPELINE', '1')                 # these are features of the threaded search
    name = 'bern_m6'
    if not os.path.exists(os.path.join(GOLDEN, name + '.npz')):
        pytest.skip('fixture not generated')
    monkeypatch.setenv('FOKL_KILL_BIC', 'check')
    g, model, betas, mtx, evs = fit_case(name)
    assert model.fit_stats['kill_tests'] > 0 and model.fit_stats['bic_from_gram'] == 0
    assert model.fit_stats['bic_gram_max_rel'] < 1e-11
    monkeypatch.setenv('FOKL_KILL_BIC', 'gram')
    g, model2, betas2, mtx2, evs2 = fit_case(name)
    assert model2.fit_stats['bic_from_gram'] == model2.fit_stats['kill_tests']
    assert np.array_equal(mtx, mtx2)
    np.testing.assert_allclose(evs2, evs, rtol=1e-11)


def test_model_saved_by_the_reference_loads_and_evaluates():
    """tests/golden/ref_saved_model.fokl was written by the reference's own ``save`` (make_golden.py saved_model): the
    product's ``load`` maps the pickled class name onto its own class; ``evaluate`` must reproduce what the reference
    returned for the reloaded model (same np.random seed -> same subset of draws, FR:934)."""
    path = os.path.join(GOLDEN, 'ref_saved_model.fokl')
    want = np.load(os.path.join(GOLDEN, 'ref_saved_model_expected.npz'))
    model = FoKLRoutines.load(path)
    assert type(model) is FoKLRoutines.FoKL
    assert np.array_equal(model.betas, want['betas']) and np.array_equal(model.mtx, want['mtx'])
    model._backend_override = OracleBackend()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        np.random.seed(6)
        mean, bounds = model.evaluate(want['inputs'], clean=True, ReturnBounds=True)
    np.testing.assert_allclose(mean, want['mean'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(bounds, want['bounds'], rtol=1e-12, atol=1e-12)
    # and the round trip through this package's save keeps working
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        again = FoKLRoutines.load(model.save('roundtrip', tmp))
    assert np.array_equal(again.betas, want['betas'])


@pytest.mark.parametrize('name', ['bern_m6', 'bern_m4_way3'])
def test_tentative_tapes_and_rewinds_change_nothing(monkeypatch, name):
    """The next kill test's noise tape is requested before the decision that the test is run and the stream is rewound
    when it is not (FOKL_TENTATIVE_TAPES: 0 = off, 1 = on, test = additionally record-and-discard a bogus tape before
    every request): draws, model, BIC trace and the final numpy RNG state must be identical in all three modes."""
    monkeypatch.setenv('FOKL_NOISE_PIPELINE', '1')                 # these are features of the threaded search
    if not os.path.exists(os.path.join(GOLDEN, name + '.npz')):
        pytest.skip('fixture not generated')
    runs = {}
    for mode in ('0', '1', 'test'):
        monkeypatch.setenv('FOKL_TENTATIVE_TAPES', mode)
        g, model, betas, mtx, evs = fit_case(name)
        runs[mode] = (betas, mtx, evs, rng_fingerprint(), dict(model.fit_stats))
    assert runs['0'][4]['tapes_rewound'] == 0
    assert runs['test'][4]['tapes_rewound'] > runs['1'][4]['tapes_rewound'] >= 0
    for mode in ('1', 'test'):
        assert np.array_equal(runs[mode][0], runs['0'][0])
        assert np.array_equal(runs[mode][1], runs['0'][1])
        assert np.array_equal(runs[mode][2], runs['0'][2])
        assert runs[mode][3] == runs['0'][3]


def test_thread_plan_follows_the_cpu_budget(monkeypatch):
    """The host pipeline's thread counts come from the CPUs the process may keep busy: affinity mask, cgroup quota,
    ranks per node; explicit FOKL_*_THREADS win."""
    monkeypatch.delenv('FOKL_CHAIN_THREADS', raising=False)
    monkeypatch.delenv('FOKL_FINISH_THREADS', raising=False)
    monkeypatch.delenv('FOKL_SPECTRAL_THREADS', raising=False)
    monkeypatch.delenv('FOKL_FINISH_LOG', raising=False)
    cases = ((16, (2, 2, 8)), (8, (1, 1, 5)), (5, (1, 1, 3)), (4, (1, 1, 2)), (3, (1, 0, 2)), (2, (1, 0, 1)), (1, (1, 0, 1)))
    for budget, plan in cases:
        monkeypatch.setattr(host_pipeline, '_cpu_budget', lambda b=budget: b)
        assert host_pipeline._thread_plan() == plan
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')                  # libm's scalar log: finishing needs three threads
    for budget, plan in ((16, (2, 3, 8)), (7.5, (1, 2, 3))):
        monkeypatch.setattr(host_pipeline, '_cpu_budget', lambda b=budget: b)
        assert host_pipeline._thread_plan() == plan
    monkeypatch.delenv('FOKL_FINISH_LOG')
    monkeypatch.setattr(host_pipeline, '_cpu_budget', lambda: 2)
    monkeypatch.setenv('FOKL_FINISH_THREADS', '5')
    assert host_pipeline._thread_plan() == (1, 5, 1)
    monkeypatch.undo()
    assert host_pipeline._cpu_budget() >= 1
    # the stream's bulk threads (csrc/fokl_stream.cpp) follow the same budget
    assert [host_pipeline._bulk_threads(b) for b in (16, 8, 5, 4, 2)] == [4, 2, 2, 1, 1]


def _search_with_pipeline(monkeypatch, draws=40):
    monkeypatch.setenv('FOKL_PIN_L3', '0')
    np.random.seed(21)
    stream = _capi.LegacyStream()
    fs = engine.ForwardSelection(OracleBackend(), 1000, 3, 5, 4, 1.0, 4, 1.0, 3, draws, draws // 2, False, False,
                                 0.05, 0.5, 2, False, stream)
    fs.host = engine.HostPipeline(stream, draws)
    return fs, stream


def test_tapes_on_order_are_used_when_the_sizes_fit_and_sent_back_when_not(monkeypatch):
    """ForwardSelection._speculate / _tape_for: orders that agree with a newer prediction stay, the rest goes back
    (youngest first), the evaluation that comes takes the oldest order if its size fits and otherwise rewinds them all;
    whatever was guessed, the stream ends exactly where the tapes that were used leave it."""
    fs, stream = _search_with_pipeline(monkeypatch)
    ref = _capi.LegacyStream()
    try:
        astar = lambda p1: (fs.a + 1 + fs.n / 2 + p1 / 2, fs.atau + (p1 - 1) / 2)
        fs._speculate([9, 8, 7, 6])
        assert [size for size, _ in fs._spec] == [9, 8, 7, 6]
        fs._speculate([9, 8, 5])                            # 7, 6 go back, 5 is placed
        assert [size for size, _ in fs._spec] == [9, 8, 5] and fs.stats['tapes_rewound'] == 2
        used = [fs._tape_for(9), fs._tape_for(8)]           # as predicted
        assert [size for size, _ in fs._spec] == [5] and not any(job.unresolved for job in used)
        used.append(fs._tape_for(4))                        # not as predicted: 5 goes back, 4 is a plain request
        assert not fs._spec and fs.stats['tapes_rewound'] == 3
        fs._speculate([4, 3])
        fs._drop_speculation()
        for job in used:
            fs.host.abandon(job)
        for p1, job in zip((9, 8, 4), used):
            want = _capi.noise_tape(p1, 40, *astar(p1), ref)
            _capi.finish_tape_blocks(want)                  # the pipeline's tapes are completed as they are recorded
            job.wait()
            assert np.array_equal(job.result.normals, want.normals) and np.array_equal(job.result.gam_tau, want.gam_tau)
    finally:
        fs.host.close()
        stream.publish()
    a, b = stream.as_numpy_state(), ref.as_numpy_state()
    assert np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def test_a_chain_started_ahead_is_adopted_or_given_up_without_losing_its_tape(monkeypatch):
    """ForwardSelection._chain_ahead: the chain of the evaluation expected next runs on its tentative tape before the
    driver gets there.  Adopted, it is the chain a plain commit would have submitted (same draws); given up -- also with
    its tape sent back under it -- nothing fails, and every buffer comes back to the spares exactly once."""
    fs, stream = _search_with_pipeline(monkeypatch)
    engine.drop_spare_buffers()
    try:
        p1 = 6
        gram = np.eye(p1 + 1) * 40.0 + 1.0
        g2 = fs.host.spectral(gram, np.arange(p1))
        other = fs.host.spectral(gram, np.arange(p1 - 1))
        g2.wait(), other.wait()
        dtd = gram[p1, p1]
        fs._speculate([p1, p1, p1 - 1])
        fs._chain_ahead(g2, p1, dtd)
        assert fs._prechain is not None and fs.stats['chains_ahead'] == 1
        fs._chain_ahead(g2, p1, dtd)                        # the same guess again: nothing new
        assert fs.stats['chains_ahead'] == 1
        pending = (g2.wait(), np.arange(p1, dtype=np.int32), None, dtd, None)
        noise_job, chain_job, w_raw = fs._commit(pending)   # adopted
        assert fs._prechain is None and fs.stats['chains_ahead_unused'] == 0 and not chain_job.ignore_failure
        w, flag = chain_job.wait()
        noise_job.wait()
        fs._chain_ahead(g2, p1, dtd)                        # second tape of that size: started, then the guess changes
        first_guess = fs._prechain
        pending = (other.wait(), np.arange(p1 - 1, dtype=np.int32), None, dtd, None)
        fs._tape_for(p1 - 1)                                # not what is on order: everything goes back, chain given up
        assert fs._prechain is None and fs.stats['chains_ahead_unused'] == 1 and not fs._spec
        first_guess[2].wait()                               # its tape was sent back under it: no exception
        assert np.all(np.isfinite(w)) and flag[0] == 0
    finally:
        fs.host.close()
        stream.publish()
    spares = [raw for stack in engine._thread_spares().values() for raw in stack]
    assert len({raw.__array_interface__['data'][0] for raw in spares}) == len(spares)        # no buffer came back twice


def _rendezvous_worker(rank, queue):
    from fokl_gpy_amd import dist as _dist
    uid, path = _dist._exchange_unique_id(rank, 2, lambda: bytes(range(128)), tag='fokl_test', timeout_s=30.0)
    queue.put((rank, uid, path))


def test_rccl_id_rendezvous_survives_a_stale_file_and_a_late_rank_zero(monkeypatch):
    """Rank 0 serves the 128-byte RCCL id over a socket whose port it advertises in a 0600 temp file named after
    MASTER_PORT / run id / restart count (no torch in a GPU process; no process ids in the name).
    Here rank 1 starts first and finds the leftover of a crashed launch (a port nobody listens on) and a planted file
    with loose permissions: it must keep polling until the live rank 0 answers."""
    import multiprocessing as mp
    import socket
    import time
    from fokl_gpy_amd import dist as _dist
    monkeypatch.setenv('MASTER_PORT', '45991')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    path = _dist._rendezvous_path('fokl_test')
    assert str(os.getppid()) not in os.path.basename(path) and str(os.getpid()) not in os.path.basename(path)
    dead = socket.socket()
    dead.bind(('127.0.0.1', 0))
    dead_port = dead.getsockname()[1]
    dead.close()
    with open(path, 'w') as fh:
        fh.write(f'{dead_port}\n')
    os.chmod(path, 0o644)                                  # loose permissions: ignored outright
    ctx = mp.get_context('fork')
    queue = ctx.Queue()
    late = ctx.Process(target=_rendezvous_worker, args=(1, queue))
    late.start()
    time.sleep(0.2)
    os.chmod(path, 0o600)                                  # now a well-formed but stale advertisement
    time.sleep(0.2)
    first = ctx.Process(target=_rendezvous_worker, args=(0, queue))
    first.start()
    got = sorted(queue.get(timeout=30) for _ in range(2))
    late.join(10)
    first.join(10)
    assert got[0][1] == got[1][1] == bytes(range(128)) and got[0][2] == got[1][2] == path
    assert not os.path.exists(path)                        # rank 0 cleans up once everybody has been served


def _random_problem(seed):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(150, 900)), int(rng.integers(1, 6))
    x = rng.random((n, m))
    y = np.sin(3 * x[:, 0]) + (x[:, 1 % m] * x[:, 2 % m] if m > 1 else 0) + 0.1 * rng.standard_normal(n)
    kw = dict(kernel=1, burnin=int(rng.integers(20, 120)), draws=int(rng.integers(20, 120)),
              tolerance=int(rng.integers(1, 4)), way3=bool(rng.integers(0, 2)), aic=bool(rng.integers(0, 2)),
              UserWarnings=False, ConsoleOutput=False)
    if rng.integers(0, 3) == 0:
        kw.update(threshstda=0.0, threshstdb=100.0, threshav=float(rng.random()))
    return x, y, kw


@pytest.mark.parametrize('seed', [3, 10, 29, 53, 70])
def test_pipelined_search_equals_the_inline_search(monkeypatch, seed):
    """Random small problems (seeds 29 and 53: single-input searches whose sub-stages end on the model of the one
    before -- the reference scores identical models identically and `ev < min(evs)` is an exact tie there): the
    threaded search in its default and in its most speculative configuration must select the same model, stop at the
    same sub-stage and leave numpy's stream where the in-line search leaves it."""
    x, y, kw = _random_problem(seed)

    def run(**env):
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = FoKLRoutines.FoKL(**kw)
            model._backend_override = OracleBackend()
            np.random.seed(seed + 7)
            betas, mtx, evs = model.fit(x, y, clean=True)
        return betas, mtx, evs, rng_fingerprint()

    ref = run(FOKL_NOISE_PIPELINE='0')
    for env in (dict(FOKL_NOISE_PIPELINE='1', FOKL_TENTATIVE_TAPES='1', FOKL_FORESIGHT='8', FOKL_KILL_BIC='auto',
                     FOKL_LOOKAHEAD='3'),
                dict(FOKL_NOISE_PIPELINE='1', FOKL_TENTATIVE_TAPES='test', FOKL_FORESIGHT='1000', FOKL_KILL_BIC='gram',
                     FOKL_LOOKAHEAD='1')):
        got = run(**env)
        assert np.array_equal(got[1], ref[1]) and got[3] == ref[3] and got[2].shape == ref[2].shape
        np.testing.assert_allclose(got[2], ref[2], rtol=1e-10)
        np.testing.assert_allclose(got[0], ref[0], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize('seed', [3, 10, 29, 41, 53, 70, 91])
def test_search_with_device_chains_equals_the_inline_search(monkeypatch, seed):
    """G3 of the kill-test candidates on a (stand-in) device engine whose chains answer milliseconds late: the search
    takes its second-clause decisions from the guessed intercept scale, confirms every one of them when the chains
    arrive, and must select the same model, score the same BIC trace, return the same draws and leave numpy's stream
    where the in-line search leaves it.  Also: a forced wrong guess is caught by the verification and the search is
    repeated on host chains; an engine without free slots hands the chains back to the host threads."""
    from helpers import StandInChainEngine
    x, y, kw = _random_problem(seed)
    made = []

    def factory(**opts):
        def make():
            made.append(StandInChainEngine(**opts))
            return made[-1]
        return make

    def run(factory_=None, **env):
        for key in ('FOKL_NOISE_PIPELINE', 'FOKL_GUESS_TEST_FLIP', 'FOKL_GUESS_MARGIN', 'FOKL_KILL_BIC'):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        monkeypatch.setattr(host_pipeline, '_chain_engine_factory', factory_)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = FoKLRoutines.FoKL(**kw)
            model._backend_override = OracleBackend()
            np.random.seed(seed + 7)
            betas, mtx, evs = model.fit(x, y, clean=True)
        return betas, mtx, evs, rng_fingerprint(), model.fit_stats, [t['cols'] for t in model.fit_trace]

    def same(got, ref):
        assert np.array_equal(got[1], ref[1]) and got[3] == ref[3] and got[2].shape == ref[2].shape and got[5] == ref[5]
        np.testing.assert_allclose(got[2], ref[2], rtol=1e-10)
        np.testing.assert_allclose(got[0], ref[0], rtol=1e-7, atol=1e-9)

    ref = run(FOKL_NOISE_PIPELINE='0')
    got = run(factory(), FOKL_NOISE_PIPELINE='1')
    same(got, ref)
    st = got[4]
    assert st['searches_repeated'] == 0 and st['guesses_verified'] == st['guessed']
    assert st['device_chains'] == made[-1].issued and made[-1].alive == 0       # every slot went back
    kills_accepted = st['device_chains']
    if st['guessed']:
        # the same search with one guess taken wrong on purpose: caught, repeated without guessing, same answer
        flipped = run(factory(), FOKL_NOISE_PIPELINE='1', FOKL_GUESS_TEST_FLIP='1')
        same(flipped, ref)
        assert flipped[4]['searches_repeated'] == 1 and flipped[4]['device_chains'] == 0 and made[-1].alive == 0
    # K3 per candidate: rejected candidates have device chains too (released without anybody reading them)
    same(run(factory(), FOKL_NOISE_PIPELINE='1', FOKL_KILL_BIC='device'), ref)
    assert made[-1].alive == 0
    # an engine with a single slot: most chains fall back to the host threads, results unchanged
    starved = run(factory(slots=1), FOKL_NOISE_PIPELINE='1')
    same(starved, ref)
    assert made[-1].alive == 0 and (kills_accepted < 2 or made[-1].refused > 0)
    # borderline proposals wait for the chain instead of guessing: a margin nothing can clear
    waited = run(factory(), FOKL_NOISE_PIPELINE='1', FOKL_GUESS_MARGIN='1e9')
    same(waited, ref)
    assert waited[4]['guessed'] == 0


def test_a_reap_between_tape_request_and_chain_cannot_recycle_the_tape(monkeypatch):
    """the recorder may be done with a tape before the chain that reads it is submitted; a _reap() in
    between (HostPipeline.spectral -> _track does one whenever 24 jobs are live) must not hand the tape's buffer
    out again -- chain() takes its output buffer from the same pool."""
    import time
    monkeypatch.setenv('FOKL_PIN_L3', '0')
    np.random.seed(3)
    stream = _capi.LegacyStream()
    engine.drop_spare_buffers()                            # spares stay with the thread from one fit to the next
    host = engine.HostPipeline(stream, 40)
    try:
        p1 = 5
        job = host.request(p1, 12.0, 6.0)
        while job.result.progress[0] < 40:
            time.sleep(0.001)
        time.sleep(0.01)
        host._reap()                                       # the noise job is done and leaves the live list ...
        assert not any(host._spare.values())               # ... but its buffer is not up for grabs
        gram = np.eye(p1 + 1) * 50.0
        gram[:p1, p1] = gram[p1, :p1] = 1.0
        spec = host.spectral(gram, np.arange(p1)).wait()
        chain_job, w_raw = host.chain(spec, 1.0, 1.0, 50.0, 0.2, 0.2, job)
        assert not np.shares_memory(w_raw, job.result.normals)
        assert job.held is None and len(chain_job.recycle) == 1
        chain_job.wait()
        host._reap()
        assert sum(len(v) for v in host._spare.values()) == 1     # now the tape buffer is back
        # a discarded tentative tape comes back as well, and the stream is where it was before it
        before = host.request(p1, 12.0, 6.0, tentative=True)
        host.discard(before)
        before.wait()
        host._reap()
        assert sum(len(v) for v in host._spare.values()) == 1 and before.held is None
    finally:
        host.close()
        stream.publish()


@pytest.mark.parametrize('name', ['bern_m3', 'bern_m6', 'bern_m4_way3'])
def test_lapack_signs_mode_follows_the_untouched_reference(monkeypatch, name):
    """FOKL_EIGH_SIGNS=lapack: eigenvectors keep the signs LAPACK returns -- the reference as shipped, whose kill tests
    then see other (equally valid) draws and select another model in 6 of the 10 fixtures.  On the host that made the
    fixture (same BLAS / LAPACK build: helpers.same_host_as) the product in that mode walks the untouched reference's
    sequence of gibbs calls past the point where the sign-canonical search leaves it (bern_m3: 18 calls against 5,
    bern_m4_way3: 14 against 6) -- until the last bits of a kill test's XtX (here a sub-block of the sub-stage's Gram
    summed in another order, FR:1676-1683 recomputes it with dgemm) flip one of LAPACK's signs: from there on it is a
    third, equally valid, realisation (bern_m6: call 25, one before the canonical search parts).  tests/golden/
    sign_sensitivity.py --last-bits shows the untouched reference parting from ITSELF under such last-bit noise, which
    is why parity is pinned on the canonical signs.  On any other host LAPACK's signs are not comparable at all and the
    test does not apply."""
    g, hy, kname, kid, phis = load_case(name)
    from helpers import same_host_as
    if not same_host_as(g):
        pytest.skip('fixture made on a host with another BLAS / LAPACK build: its eigenvector signs are not this host\'s')
    monkeypatch.setenv('FOKL_EIGH_SIGNS', 'lapack')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = make_model(kname, phis, hy)
        np.random.seed(int(g['seed']))
        betas, mtx, evs = model.fit(g['raw_inputs'], g['raw_data'], clean=True)
    sizes = [t['cols'] for t in model.fit_trace]
    ref = [int(v) for v in g['ref_gibbs_sizes']]
    canon = [int(v) for v in g['canon_gibbs_sizes']]

    def together(a, b):
        return next((i for i, (u, v) in enumerate(zip(a, b)) if u != v), min(len(a), len(b)))

    with_ref, canon_with_ref = together(sizes, ref), together(canon, ref)
    assert canon_with_ref < len(ref), "this fixture's searches do not part: pick another"
    assert together(sizes, canon) <= canon_with_ref, 'the mode changed nothing'
    if canon_with_ref < 10:
        assert with_ref > canon_with_ref + 5, (with_ref, canon_with_ref)
    else:
        assert with_ref >= 10, (with_ref, canon_with_ref)
    # the sub-stages both completed before parting carry the reference's BIC
    shared = 0
    while shared < min(len(evs), len(g['ref_evs'])) and abs(evs[shared] - g['ref_evs'][shared]) <= 1e-9 * abs(g['ref_evs'][shared]):
        shared += 1
    canon_shared = 0
    while (canon_shared < min(len(g['canon_evs']), len(g['ref_evs'])) and
           abs(g['canon_evs'][canon_shared] - g['ref_evs'][canon_shared]) <= 1e-9 * abs(g['ref_evs'][canon_shared])):
        canon_shared += 1
    assert shared >= canon_shared


def test_lookahead_knobs_are_bounded_to_what_was_verified():
    """FOKL_LOOKAHEAD / FOKL_LOOKAHEAD_DERIVED deeper than engine.LOOKAHEAD_MAX (the deepest window every BASELINE
    configuration has been run with) are brought back to it with a warning: 48 on configs[3] used to stop the fit with
    "the pre-state of a segment is not (or no longer) in the ring"."""
    import warnings as _warnings
    from fokl_gpy_amd import engine as _engine
    with _warnings.catch_warnings(record=True) as seen:
        _warnings.simplefilter('always')
        assert _engine._bounded_lookahead('FOKL_LOOKAHEAD', '48') == _engine.LOOKAHEAD_MAX == 24
    assert len(seen) == 1 and issubclass(seen[0].category, RuntimeWarning) and 'FOKL_LOOKAHEAD=48' in str(seen[0].message)
    with _warnings.catch_warnings(record=True) as seen:
        _warnings.simplefilter('always')
        assert [_engine._bounded_lookahead('FOKL_LOOKAHEAD', v) for v in ('0', '12', '24', '-5')] == [0, 12, 24, 0]
    assert not seen


def test_matrix_free_residual_pass_is_offered_only_inside_its_slot_layouts():
    """engine.HipBackend.resid_terms_supported mirrors fokl_bic_resid_terms_launch's limits (include/fokl_hip.h): one or
    two inputs per term, the distinct (input, order) factors inside one of the layouts inputs x orders-per-input
    8 x 1, 16 x 1, 8 x 2, 4 x 4, 2 x 8 (16 slots), 8 x 4, 16 x 2, 4 x 8 (32); spline factors pay only from 3.5 model
    columns per slot on."""
    import types
    from fokl_gpy_amd import engine
    shape = staticmethod(engine.HipBackend._factor_shape)
    bern, spl = types.SimpleNamespace(kernel_id=1, _factor_shape=shape), types.SimpleNamespace(kernel_id=0, _factor_shape=shape)
    ok = lambda be, t: engine.HipBackend.resid_terms_supported(be, np.asarray(t, dtype=np.int32))
    pay = lambda be, t: engine.HipBackend.resid_terms_pay_from(be, np.asarray(t, dtype=np.int32))
    mains = np.eye(8, dtype=np.int32)
    pairs = engine.distinct_arrangements([1, 1] + [0] * 6)
    assert ok(bern, mains) and ok(bern, np.vstack([mains, pairs]))                      # 8 x 1
    second = np.vstack([mains, 2 * mains, engine.distinct_arrangements([2, 1] + [0] * 6)])
    assert ok(bern, second)                                                              # 8 x 2
    assert ok(bern, np.vstack([second, 3 * mains, 4 * mains]))                           # 8 x 4 (32 slots)
    assert not ok(bern, np.vstack([second, 3 * mains, 4 * mains, 5 * mains]))            # 8 inputs x 5 orders: no layout
    wide = np.eye(16, dtype=np.int32)
    assert ok(bern, np.vstack([wide, 2 * wide])) and not ok(bern, np.vstack([wide, 2 * wide, 3 * wide]))   # 16 x 2
    assert ok(bern, np.vstack([k * np.eye(4, 8, dtype=np.int32) for k in (1, 2, 3, 4)]))  # 4 x 4
    assert ok(bern, np.eye(16, dtype=np.int32)) and not ok(bern, np.eye(17, dtype=np.int32))
    three_way = np.zeros((1, 8), dtype=np.int32)
    three_way[0, :3] = 1
    assert not ok(bern, three_way) and not ok(bern, np.zeros((1, 8)))
    assert not ok(bern, 9 * mains[:1]) and ok(spl, 9 * mains[:1])                        # Bernoulli orders stop at 8
    assert pay(bern, second) == 0
    assert pay(spl, mains) == 28 and pay(spl, second) == 56 and pay(spl, np.eye(16, dtype=np.int32)) == 56
