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

from helpers import GOLDEN, OracleBackend
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
    assert np.array_equal(mtx, G[pre + 'mtx']) and built == bool(G[pre + 'built'])
    assert np.array_equal(evs, G[pre + 'evs']) and np.array_equal(betas, G[pre + 'betas'])
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
    and raises TypeError in the others; a built model is refused (cases 2 / 3 are not part of this build)."""
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
        built = FoKLRoutines.FoKL(built=True, **kw)
        built._backend_override = OracleBackend()
        with pytest.raises(NotImplementedError):
            built.fit(x, y, clean=True)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', CASES)
def test_first_update_call_on_gpu(tag):
    check_against_reference(tag, *fit_product(tag))
