"""N3: ``fit(update=True)`` / ``fitupdate`` on a model without a prior -- gibbs_Xin_update "case 1" under
the fitupdate driver (FR:1850-2152, 2473-2583).  Fixtures: tests/golden/fitupdate.npz, produced by the real reference
(make_golden.py case_fitupdate), untouched and with the sign-canonical eigh.

Tolerances: interaction matrix, number of sub-stages, `built` flag and numpy's stream after the fit exact; BIC trace
1e-9 relative; draws 1e-9 of the column scale against the sign-canonical variant (the oracle reproduces both variants
bit for bit on this image)."""
import os
import warnings

import numpy as np
import pytest

from helpers import GOLDEN, OracleBackend, same_host_as
from fokl_gpy_amd import FoKLRoutines, getKernels
from oracle import fokl_oracle as O

G = np.load(os.path.join(GOLDEN, 'fitupdate.npz'), allow_pickle=False)
CASES = [str(c) for c in G['cases']]


def case_setup(tag):
    kern = str(G[f'{tag}_kernel'])
    hy = {str(k): float(v) for k, v in zip(G[f'{tag}_hyper_keys'], G[f'{tag}_hyper_vals'])}
    for k in ('burnin', 'draws', 'tolerance'):
        if k in hy:
            hy[k] = int(hy[k])
    for k in ('gimmie', 'aic'):
        if k in hy:
            hy[k] = bool(hy[k])
    for k in ('a', 'atau'):
        if k in hy and float(hy[k]).is_integer():
            hy[k] = int(hy[k])
    fit_keys = {str(k) for k in G[f'{tag}_fit_keys']}
    init = {k: v for k, v in hy.items() if k not in fit_keys}
    fit_kw = {k: v for k, v in hy.items() if k in fit_keys}
    if kern == 'Cubic Splines':
        phis, kid = getKernels.table_to_phis(np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table']), O.KERNEL_SPLINES
    else:
        phis, kid = getKernels.bernoulli(), O.KERNEL_BERNOULLI
    return kern, phis, kid, init, fit_kw, hy


def loose_equal(evs, want_evs, betas, want_betas):
    """SURVEY 8(c)'s host-independent tolerances (what the product is held to): BIC 1e-9 relative, draws 1e-9 of the
    column scale.  Used where this host's BLAS / LAPACK is not the one that made the fixture; there the last bits of
    the reference are not reproducible by the reference itself."""
    evs, want_evs = np.ravel(evs), np.ravel(want_evs)
    assert evs.shape == want_evs.shape and np.max(np.abs(evs - want_evs) / np.abs(want_evs)) < 1e-9
    assert betas.shape == want_betas.shape
    both_nan = np.isnan(betas) & np.isnan(want_betas)
    scale = np.nanmax(np.abs(want_betas), axis=0)
    assert np.array_equal(np.isnan(betas), np.isnan(want_betas))
    assert np.max(np.where(both_nan, 0.0, np.abs(betas - want_betas) / scale)) < 1e-9


def rng_fingerprint():
    import hashlib
    st = np.random.get_state()
    h = hashlib.sha256(st[1].tobytes()).hexdigest()[:16]
    return np.array([int(h, 16) % (2 ** 53), st[2], st[3]], dtype=np.float64), float(st[4])


@pytest.mark.parametrize('tag', CASES)
@pytest.mark.parametrize('variant', ['ref', 'canon'])
def test_oracle_restatement_is_the_reference_bit_for_bit(tag, variant):
    kern, phis, kid, init, fit_kw, hy = case_setup(tag)
    sig0 = hy.pop('sigsqd0', 0.5)
    np.random.seed(int(G[f'{tag}_seed']))
    betas, mtx, evs, built = O.fitupdate_first(G[f'{tag}_norm_inputs'], G[f'{tag}_norm_data'], phis, kid,
                                               eigh=O.eigh_canonical if variant == 'canon' else O.eigh_reference,
                                               sigsqd0=sig0, **hy)
    fp, cache = rng_fingerprint()
    pre = f'{tag}_{variant}_'
    if not same_host_as(G):
        if variant == 'ref':
            pytest.skip('the untouched reference is reproducible only on the host whose LAPACK made the fixture')
        loose_equal(evs, G[pre + 'evs'], betas, G[pre + 'betas'])
    else:
        assert np.array_equal(evs, G[pre + 'evs']) and np.array_equal(betas, G[pre + 'betas'])
    assert np.array_equal(mtx, G[pre + 'mtx']) and built == bool(G[pre + 'built'])
    assert np.array_equal(fp, G[pre + 'rng']) and cache == float(G[pre + 'rng_cache'])


def fit_product(tag, backend=None):
    kern, phis, kid, init, fit_kw, _ = case_setup(tag)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=kern, phis=phis, update=True, UserWarnings=False, ConsoleOutput=False, **init)
        if backend is not None:
            model._backend_override = backend
        np.random.seed(int(G[f'{tag}_seed']))
        betas, mtx, evs = model.fit(G[f'{tag}_raw_inputs'], G[f'{tag}_raw_data'], clean=True, **fit_kw)
    return model, betas, mtx, evs, rng_fingerprint()


def check_against_reference(tag, model, betas, mtx, evs, fp):
    pre = f'{tag}_canon_'
    assert np.array_equal(model.inputs, G[f'{tag}_norm_inputs'])
    assert mtx.shape == G[pre + 'mtx'].shape and np.array_equal(mtx, G[pre + 'mtx'])
    assert np.array_equal(mtx, G[f'{tag}_ref_mtx'])                       # the untouched reference selects the same model
    assert model.built == bool(G[pre + 'built'])
    assert len(evs) == len(G[pre + 'evs']) and np.max(np.abs(evs - G[pre + 'evs']) / np.abs(G[pre + 'evs'])) < 1e-9
    assert betas.shape == G[pre + 'betas'].shape                          # all burnin + draws rows, as the reference returns
    scale = np.max(np.abs(G[pre + 'betas']), axis=0)
    assert np.max(np.abs(betas - G[pre + 'betas']) / scale) < 1e-9
    assert np.array_equal(fp[0], G[pre + 'rng']) and fp[1] == float(G[pre + 'rng_cache'])
    assert model.betas is betas and model.mtx is mtx


@pytest.mark.parametrize('tag', CASES)
def test_first_update_call_on_the_checker_backend(tag):
    check_against_reference(tag, *fit_product(tag, OracleBackend()))


def test_update_quirks_of_the_reference():
    """One input dies in the reference's shape handling (FR:2528); relats_in excludes nothing in the variants that run
    and raises TypeError in the others."""
    rng = np.random.default_rng(0)
    x, y = rng.random((80, 3)), rng.random(80)
    kw = dict(kernel=1, update=True, burnin=10, draws=10, UserWarnings=False, ConsoleOutput=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        one = FoKLRoutines.FoKL(**kw)
        one._backend_override = OracleBackend()
        with pytest.raises(ValueError):
            one.fit(x[:, :1], y, clean=True)
        results = []
        for relats in ([], np.array([[1, 0, 0]]), [1, 1, 1], [[1, 1, 0]]):
            model = FoKLRoutines.FoKL(relats_in=relats, **kw)
            model._backend_override = OracleBackend()
            np.random.seed(1)
            results.append(model.fit(x, y, clean=True))
        for _, mtx, evs in results[1:]:
            assert np.array_equal(mtx, results[0][1]) and np.array_equal(evs, results[0][2])
        for relats in (np.array([[0, 0, 1], [1, 1, 0]]), [0, 1, 1]):
            model = FoKLRoutines.FoKL(relats_in=relats, **kw)
            model._backend_override = OracleBackend()
            with pytest.raises(TypeError):
                model.fit(x, y, clean=True)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', CASES)
def test_first_update_call_on_gpu(tag):
    check_against_reference(tag, *fit_product(tag))


# ---------------------------------------------------------------------------------------------------------
# second call: priors from the first fit's posterior (gibbs_Xin_update cases 2 and 3)
# ---------------------------------------------------------------------------------------------------------

S = np.load(os.path.join(GOLDEN, 'fitupdate_sequence.npz'), allow_pickle=False)
SEQ = [str(c) for c in S['cases']]


def seq_setup(tag):
    kern = str(S[f'{tag}_kernel'])
    hy = {str(k): float(v) for k, v in zip(S[f'{tag}_hyper_keys'], S[f'{tag}_hyper_vals'])}
    for k in ('burnin', 'draws', 'tolerance', 'burn'):
        if k in hy:
            hy[k] = int(hy[k])
    for k in ('gimmie', 'aic'):
        if k in hy:
            hy[k] = bool(hy[k])
    if kern == 'Cubic Splines':
        phis, kid = getKernels.table_to_phis(np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table']), O.KERNEL_SPLINES
    else:
        phis, kid = getKernels.bernoulli(), O.KERNEL_BERNOULLI
    return kern, phis, kid, hy


@pytest.mark.parametrize('tag', SEQ)
@pytest.mark.parametrize('variant', ['ref', 'canon'])
def test_oracle_restates_the_second_update_call_bit_for_bit(tag, variant):
    """From the reference's own first-call draws and the stream position it left: cases 2 / 3 of the oracle return the
    reference's second-call numbers exactly (b / btau persist from the first call, FR:1322-1348 only fill None)."""
    kern, phis, kid, hy = seq_setup(tag)
    pre = f'{tag}_{variant}_'
    burn, sig0 = hy.pop('burn'), hy.pop('sigsqd0', 0.5)
    # replay the first call to put numpy's stream where the second call starts
    x1, y1 = S[f'{tag}_raw_inputs1'], S[f'{tag}_raw_data1']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        first = FoKLRoutines.FoKL(kernel=kern, phis=phis, UserWarnings=False, ConsoleOutput=False)
        first.clean(x1, y1, _setattr=True)
    eigh = O.eigh_canonical if variant == 'canon' else O.eigh_reference
    np.random.seed(int(S[f'{tag}_seed']))
    b1, m1, e1, built = O.fitupdate_first(first.inputs, first.data, phis, kid, eigh=eigh, sigsqd0=sig0, **hy)
    strict = same_host_as(S)
    if not strict and variant == 'ref':
        pytest.skip('the untouched reference is reproducible only on the host whose LAPACK made the fixture')
    if strict:
        assert np.array_equal(b1, S[pre + 'betas1'])
    else:
        loose_equal(e1, S[pre + 'evs1'], b1, S[pre + 'betas1'])
        b1 = S[pre + 'betas1']                      # the second call's prior is the REFERENCE's first-call draws
    assert built == bool(S[pre + 'built1'])
    assert np.array_equal(rng_fingerprint()[0], S[pre + 'rng_mid'])
    b2, m2, e2, _ = O.fitupdate_next(S[f'{tag}_norm_inputs2'], S[f'{tag}_norm_data2'], phis, kid, b1, burn=burn, eigh=eigh,
                                     sigsqd0=sig0, **dict(hy, b=float(S[f'{tag}_b']), btau=float(S[f'{tag}_btau'])))
    fp, cache = rng_fingerprint()
    assert np.array_equal(m2, S[pre + 'mtx2'])
    if strict:
        assert np.array_equal(e2, S[pre + 'evs2']) and np.array_equal(np.asarray(b2), S[pre + 'betas2'])
    else:
        loose_equal(e2, S[pre + 'evs2'], np.asarray(b2), S[pre + 'betas2'])
    assert np.array_equal(fp, S[pre + 'rng']) and cache == float(S[pre + 'rng_cache'])


def update_twice(tag, backend_factory=None):
    kern, phis, kid, hy = seq_setup(tag)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=kern, phis=phis, update=True, UserWarnings=False, ConsoleOutput=False, **hy)
        if backend_factory is not None:
            model._backend_override = backend_factory()
        np.random.seed(int(S[f'{tag}_seed']))
        b1, m1, e1 = model.fit(S[f'{tag}_raw_inputs1'], S[f'{tag}_raw_data1'], clean=True)
        mid = rng_fingerprint()
        b2, m2, e2 = model.fit(S[f'{tag}_raw_inputs2'], S[f'{tag}_raw_data2'], clean=True)
    return model, (b1, m1, e1, mid), (b2, m2, e2, rng_fingerprint())


def check_sequence(tag, model, first, second):
    """Same tolerances as everywhere: models, shapes and numpy's stream exact, BIC 1e-9 relative, draws 1e-9 of the
    column scale -- although the second call inverts the covariance of the first call's draws (condition numbers
    5e3 .. 9e4 in these fixtures) the measured differences stay at 1e-12 / 6e-11."""
    pre = f'{tag}_canon_'
    b1, m1, e1, mid = first
    b2, m2, e2, end = second
    assert np.array_equal(m1, S[pre + 'mtx1']) and model.built == bool(S[pre + 'built1'])
    assert np.max(np.abs(b1 - S[pre + 'betas1']) / np.max(np.abs(S[pre + 'betas1']), axis=0)) < 1e-9
    assert np.array_equal(mid[0], S[pre + 'rng_mid']) and mid[1] == float(S[pre + 'rng_mid_cache'])
    assert np.array_equal(model.inputs, S[f'{tag}_norm_inputs2'])        # second batch normalised with the first's min / max
    assert abs(model.b - float(S[f'{tag}_b'])) <= 1e-15 * abs(float(S[f'{tag}_b']))
    assert np.array_equal(m2, S[pre + 'mtx2']) and np.array_equal(m2, S[f'{tag}_ref_mtx2'])
    assert e2.shape == S[pre + 'evs2'].shape                              # [k, 1], as the reference leaves it
    assert np.max(np.abs(e2 - S[pre + 'evs2']) / np.abs(S[pre + 'evs2'])) < 1e-9
    scale = np.max(np.abs(S[pre + 'betas2']), axis=0)
    assert b2.shape == S[pre + 'betas2'].shape and np.max(np.abs(b2 - S[pre + 'betas2']) / scale) < 1e-9
    assert np.array_equal(end[0], S[pre + 'rng']) and end[1] == float(S[pre + 'rng_cache'])


@pytest.mark.parametrize('tag', SEQ)
def test_second_update_call_on_the_checker_backend(tag):
    check_sequence(tag, *update_twice(tag, OracleBackend))


@pytest.mark.gpu
@pytest.mark.parametrize('tag', SEQ)
def test_second_update_call_on_gpu(tag):
    check_sequence(tag, *update_twice(tag))
