"""The native search core (csrc/fokl_search.cpp: tapes on order, G2 ahead, chains, the kill-test loop FR:1666-1690) against
engine.py's own statement of the same logic (FOKL_SEARCH=python), on the CPU stand-in backend: same sequence of model
evaluations (sizes, BICs, kill flags), same selected model, same draws, same consumption of numpy's random stream."""
import warnings

import numpy as np
import pytest

from helpers import OracleBackend, load_case
from fokl_gpy_amd import FoKLRoutines, engine, _capi


def _fit(monkeypatch, driver, name=None, problem=None, env=(), **hypers):
    monkeypatch.setenv('FOKL_SEARCH', driver)
    for key, value in env:
        monkeypatch.setenv(key, value)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if name is not None:
            g, hy, kname, kid, phis = load_case(name)
            model = FoKLRoutines.FoKL(kernel=kname, phis=phis, UserWarnings=False, ConsoleOutput=False, **{**hy, **hypers})
            x, y, seed = g['raw_inputs'], g['raw_data'], int(g['seed'])
        else:
            x, y, seed = problem
            model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False, **hypers)
        model._backend_override = OracleBackend()
        np.random.seed(seed)
        betas, mtx, evs = model.fit(x, y, clean=True)
    st = np.random.get_state()
    return model, betas, mtx, evs, (st[1].copy(), st[2], st[3], st[4])


def _same(a, b):
    ma, ba, xa, ea, sa = a
    mb, bb, xb, eb, sb = b
    assert ma.fit_stats['search_driver'] == 'native' and mb.fit_stats['search_driver'] == 'python'
    assert [(t['cols'], t['built'], t['kill']) for t in ma.fit_trace] == \
           [(t['cols'], t['built'], t['kill']) for t in mb.fit_trace]
    assert np.allclose([t['ev'] for t in ma.fit_trace], [t['ev'] for t in mb.fit_trace], rtol=1e-12, atol=0)
    assert np.array_equal(xa, xb)
    assert np.allclose(ea, eb, rtol=1e-12, atol=0)
    assert ba.shape == bb.shape and np.max(np.abs(ba - bb)) <= 1e-12 * np.max(np.abs(bb))
    assert np.array_equal(sa[0], sb[0]) and sa[1:] == sb[1:]             # the stream ends on the same state
    for key in ('gibbs_calls', 'kill_tests', 'terms_logical', 'substages'):
        assert ma.fit_stats[key] == mb.fit_stats[key], key


@pytest.mark.parametrize('name', ['bern_m1', 'bern_m3', 'bern_m4_way3', 'bern_m6', 'bern_m8_capped', 'splines_m4'])
def test_native_search_equals_the_python_search_on_the_fixtures(monkeypatch, name):
    _same(_fit(monkeypatch, 'native', name), _fit(monkeypatch, 'python', name))


def _random_problem(seed, n=600, m=5):
    rng = np.random.default_rng(seed)
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.3 * x[:, 3] ** 2 + 0.05 * rng.standard_normal(n)
    return x, y, 1000 + seed


@pytest.mark.parametrize('seed', [3, 10, 29, 53])
@pytest.mark.parametrize('env', [(), (('FOKL_TENTATIVE_TAPES', 'test'),), (('FOKL_LOOKAHEAD', '0'), ('FOKL_FORESIGHT', '0')),
                                 (('FOKL_SPECULATION', '2'),), (('FOKL_FINISH_THREADS', '0'),),
                                 (('FOKL_KILL_DECIDE', 'g2'),), (('FOKL_G2_DEFER_FROM', '4'),),
                                 (('FOKL_SPECULATE_ACROSS', '0'),), (('FOKL_KILL_DECIDE_TOL', '3e-3'),)])
def test_native_search_equals_the_python_search_whatever_is_ordered_ahead(monkeypatch, seed, env):
    """Forced rewinds before every order, no look-ahead at all, a short order book, no finish threads, kill tests decided
    from G2 of every trial model (round 4) instead of from the downdated least-squares model, G2 of accepted models
    requested only when something needs it, no tapes ordered across the sub-stage boundary: what is prepared ahead
    differs, what is evaluated does not.  FOKL_KILL_DECIDE_TOL = 3e-3: a band around the BIC to beat so wide that most
    direct decisions count as too close to call and are taken from the trial model's eigenpairs instead."""
    problem = _random_problem(seed)
    hy = dict(draws=60, burnin=60)
    a = _fit(monkeypatch, 'native', problem=problem, env=env, **hy)
    _same(a, _fit(monkeypatch, 'python', problem=problem, env=env, **hy))
    if ('FOKL_KILL_DECIDE_TOL', '3e-3') in env:
        assert a[0].fit_stats['direct_in_band'] > 0
    elif not env:
        assert a[0].fit_stats['direct_in_band'] == 0        # default band (1e-9 relative): never met on these problems


def test_native_objects_are_released(monkeypatch):
    """A fit leaves no tape on order and nothing in flight: the pool can be torn down and the stream written back (the
    state compared above); spare buffers are reused by the next fit (same process, no growth in what is kept)."""
    problem = _random_problem(7)
    a = _fit(monkeypatch, 'native', problem=problem, draws=40, burnin=40)
    b = _fit(monkeypatch, 'native', problem=problem, draws=40, burnin=40)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
    assert a[0].fit_stats['gibbs_calls'] == b[0].fit_stats['gibbs_calls']


def test_a_callback_that_raises_surfaces_after_the_native_loop(monkeypatch):
    """The device stand-in fails in the middle of a search -- inside a callback of the native sub-stage loop
    (csrc/fokl_run.cpp), or inside this file's loop (FOKL_SUBSTAGE_LOOP=python): the exception surfaces once the native
    frames have returned, and the next fit in the process works: nothing was left half torn down."""
    problem = _random_problem(3)
    boom = RuntimeError('device gone')
    real = OracleBackend.build_terms
    for loop in ('native', 'python'):
        monkeypatch.setenv('FOKL_SUBSTAGE_LOOP', loop)
        calls = []

        def failing(self, terms, slots, calls=calls):
            calls.append(1)
            if len(calls) == 3:
                raise boom
            return real(self, terms, slots)

        monkeypatch.setattr(OracleBackend, 'build_terms', failing)
        with pytest.raises(RuntimeError, match='device gone'):
            _fit(monkeypatch, 'native', problem=problem, draws=40, burnin=40)
        monkeypatch.setattr(OracleBackend, 'build_terms', real)
        _fit(monkeypatch, 'native', problem=problem, draws=40, burnin=40)


@pytest.mark.parametrize('name', ['bern_m3', 'bern_m4_way3', 'bern_m8_capped', 'splines_m4'])
def test_native_substage_loop_equals_the_python_one(monkeypatch, name):
    """csrc/fokl_run.cpp against engine.ForwardSelection._run, both on the native search: the same sequence of evaluations,
    model, draws, statistics of every sub-stage and end state of numpy's stream."""
    monkeypatch.setenv('FOKL_SUBSTAGE_LOOP', 'native')
    a = _fit(monkeypatch, 'native', name)
    monkeypatch.setenv('FOKL_SUBSTAGE_LOOP', 'python')
    b = _fit(monkeypatch, 'native', name)
    assert a[0].fit_stats.get('substage_loop') == 'native' and b[0].fit_stats.get('substage_loop') != 'native'
    assert [(t['cols'], t['built'], t['kill']) for t in a[0].fit_trace] == [(t['cols'], t['built'], t['kill']) for t in b[0].fit_trace]
    assert np.array_equal([t['ev'] for t in a[0].fit_trace], [t['ev'] for t in b[0].fit_trace])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[4][0], b[4][0]) and a[4][1:] == b[4][1:]
    assert len(a[0].fit_substage_stats) == len(b[0].fit_substage_stats)
    for x, y in zip(a[0].fit_substage_stats, b[0].fit_substage_stats):
        assert np.array_equal(x['mean_abs'], y['mean_abs']) and np.array_equal(x['rel_std'], y['rel_std'])
