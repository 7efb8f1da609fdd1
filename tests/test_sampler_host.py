"""Host half of the sampler (C++ in libfokl_hip.so): numpy-legacy random stream and the eigenbasis Gibbs chain."""
import numpy as np
import pytest

from fokl_gpy_amd import _capi
from oracle import fokl_oracle as O


@pytest.mark.parametrize('seed', [0, 7, 102823, 102923, 2 ** 32 - 1])
def test_stream_is_bit_identical_to_numpy(seed):
    np.random.seed(seed)
    st = _capi.LegacyStream()
    want, got = [], []
    for it in range(200):
        p = 1 + it % 11
        want.append(np.random.normal(loc=0, scale=1, size=(p, 1)).ravel())
        got.append(st.normals(p))
        shape = 4 + 1 + 500 / 2 + p / 2 + 0.37 * it              # astar-like: large shapes
        want.append(np.atleast_1d(np.random.gamma(shape, 1 / (3.0 + it))))
        got.append(st.gammas(shape, 1 / (3.0 + it), 1))
        want.append(np.atleast_1d(np.random.gamma(4 + p / 2, 0.25)))  # atau_star-like
        got.append(st.gammas(4 + p / 2, 0.25, 1))
    assert np.array_equal(np.concatenate(want), np.concatenate(got))
    a, b = np.random.get_state(), st.as_numpy_state()
    assert np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def test_small_shape_branches_match_numpy():
    np.random.seed(3)
    st = _capi.LegacyStream()
    for shape in (0.05, 0.3, 0.999, 1.0, 1.0001, 0.0):
        want = np.random.gamma(shape, 2.0, size=50)
        got = st.gammas(shape, 2.0, 50)
        assert np.array_equal(want, got), shape


def test_publish_continues_numpys_global_stream():
    np.random.seed(11)
    ref = np.random.normal(size=7)
    ref_next = np.random.normal(size=5)
    np.random.seed(11)
    st = _capi.LegacyStream()
    assert np.array_equal(st.normals(7), ref)
    st.publish()
    assert np.array_equal(np.random.normal(size=5), ref_next)


def _chain_case(seed, n, p1):
    rng = np.random.default_rng(seed)
    X = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, p1 - 1))], axis=1)
    beta = rng.standard_normal(p1)
    y = (X @ beta + 0.1 * rng.standard_normal(n))[:, None]
    return X, y


@pytest.mark.parametrize('seed,n,p1', [(1, 200, 4), (2, 500, 12), (3, 300, 1)])
def test_chain_matches_the_reference_loop(seed, n, p1):
    """Eigenbasis chain == the beta-space loop of FoKLRoutines.py:1519-1548 (restated in the oracle) to rounding."""
    X, y = _chain_case(seed, n, p1)
    a, atau = 4, 4
    b, btau = O.default_b_btau(y, a, atau)
    draws = 120
    dtd = np.transpose(y).dot(y)
    np.random.seed(100 + seed)
    res = O.gibbs(X[:, :1] * 0 + X[:, :1], y, None, O.KERNEL_BERNOULLI, X, np.zeros((p1 - 1, 1)), a, b, atau, btau,
                  draws, None, None, b / (1 + a), btau / (1 + atau), dtd, eigh=O.eigh_canonical) if p1 > 1 else \
        O.gibbs(X, y, None, O.KERNEL_BERNOULLI, X, np.zeros((0, 1)), a, b, atau, btau, draws, None, None,
                b / (1 + a), btau / (1 + atau), dtd, eigh=O.eigh_canonical)
    state_after_ref = np.random.get_state()

    np.random.seed(100 + seed)
    st = _capi.LegacyStream()
    lamb, Q = O.eigh_canonical(X.T @ X)
    qty = Q.T @ (X.T @ y)[:, 0]
    astar = a + 1 + n / 2 + p1 / 2
    atau_star = atau + (p1 - 1) / 2
    w, sigs, taus = _capi.gibbs_chain(lamb, qty, astar, atau_star, b, btau, float(dtd[0, 0]), b / (1 + a),
                                      btau / (1 + atau), draws, st, want_sig_tau=True)
    betas = w @ Q.T
    scale = np.max(np.abs(res.betas), axis=0)
    assert np.max(np.abs(betas - res.betas) / scale) < 1e-10
    assert np.allclose(sigs, res.sigs[:, 0], rtol=1e-10, atol=0)
    assert np.allclose(taus, res.taus[:, 0], rtol=1e-10, atol=0)
    mine = st.as_numpy_state()
    assert np.array_equal(mine[1], state_after_ref[1]) and mine[2:4] == state_after_ref[2:4]


def test_negative_bstar_skips_the_gamma_draw_and_goes_nan():
    """FR:1538-1541: bstar < 0 -> sigsqd = nan without consuming the stream for that draw."""
    np.random.seed(5)
    st = _capi.LegacyStream()
    lamb = np.array([1.0, 2.0])
    qty = np.array([0.5, -0.25])
    w, sigs, taus = _capi.gibbs_chain(lamb, qty, 10.0, 5.0, -1e9, 1.0, 0.0, 1.0, 1.0, 3, st, want_sig_tau=True)
    assert np.isnan(sigs[0]) and np.all(np.isnan(taus))
    # consumption: 2 normals, (no gamma), one gamma(atau_star) for the first iteration
    np.random.seed(5)
    np.random.normal(size=2)
    np.random.gamma(5.0, 1.0)
    np.random.normal(size=2)
    assert np.isfinite(w[0]).all()


def test_bad_arguments_are_rejected():
    st = _capi.LegacyStream()
    with pytest.raises(_capi.FoklNativeError):
        _capi.gibbs_chain(np.ones(2), np.ones(2), -1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 2, st)
    st.pos.value = 9999
    with pytest.raises(_capi.FoklNativeError):
        st.normals(3)
