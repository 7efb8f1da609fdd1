"""N2: bss_derivatives -- oracle pinned to the reference, host logic on CPU, HIP path on the GPU."""
import os
import warnings

import numpy as np
import pytest

from helpers import GOLDEN, OracleBackend
from fokl_gpy_amd import FoKLRoutines, getKernels
from oracle import fokl_oracle as O

G = np.load(os.path.join(GOLDEN, 'derivatives.npz'))
SPL = getKernels.table_to_phis(np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table'])
KINDS = {'bern': ('Bernoulli Polynomials', O.KERNEL_BERNOULLI, getKernels.bernoulli()),
         'spl': ('Cubic Splines', O.KERNEL_SPLINES, SPL)}


def model_from_fixture(tag, backend=None):
    kname, kid, phis = KINDS[tag]
    model = FoKLRoutines.FoKL(kernel=kname, phis=phis, burnin=40, draws=40, UserWarnings=False, ConsoleOutput=False)
    model.inputs = G[tag + '_inputs']
    model.betas = G[tag + '_betas']
    model.mtx = G[tag + '_mtx']
    model.minmax = [list(r) for r in G[tag + '_minmax']]
    if backend is not None:
        model._backend_override = backend
    return model


@pytest.mark.parametrize('tag', ['bern', 'spl'])
def test_oracle_derivatives_reproduce_reference(tag):
    kname, kid, phis = KINDS[tag]
    minmax = [list(r) for r in G[tag + '_minmax']]
    dy = O.bss_derivatives(G[tag + '_inputs'], G[tag + '_betas'], G[tag + '_mtx'], phis, kid, minmax,
                           [True] * 3, [True] * 3, 40)
    assert np.array_equal(dy, G[tag + '_full_draws'])
    assert np.allclose(np.mean(dy, axis=3), G[tag + '_full'], rtol=1e-13, atol=1e-15)


def check_all_forms(model, tag, tol):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        scale = np.max(np.abs(G[tag + '_full_draws']))
        for key, kwargs in (('_default', {}), ('_full', dict(d1=True, d2=True, ReturnFullArray=True)),
                            ('_full_draws', dict(d1=True, d2=True, ReturnFullArray=True, IndividualDraws=True)),
                            ('_mixed', dict(d1=1, d2=[1, 0, 0])),
                            ('_mixed_draws', dict(d1=[0, 0, 1], d2='on', IndividualDraws=True))):
            got = model.bss_derivatives(**kwargs)
            want = G[tag + key]
            assert got.shape == want.shape, (key, got.shape, want.shape)
            assert np.max(np.abs(got - want)) <= tol * scale, key
        dy, basis = model.bss_derivatives(d1=0, ReturnBasis=True)
        assert np.max(np.abs(dy - G[tag + '_basis_dy'])) <= tol * scale
        assert np.max(np.abs(basis - G[tag + '_basis'])) <= 1e-12


@pytest.mark.parametrize('tag', ['bern', 'spl'])
def test_host_logic_of_bss_derivatives(tag):
    check_all_forms(model_from_fixture(tag, OracleBackend()), tag, 1e-12)


def test_keyword_handling_matches_reference():
    model = model_from_fixture('bern', OracleBackend())
    with pytest.raises(ValueError, match="Unexpected keyword"):
        model.bss_derivatives(order=2)
    with pytest.raises(ValueError, match="must be of equal length"):
        model.bss_derivatives(d1=[1, 0])
    with pytest.raises(ValueError, match="limited to an integer"):
        model.bss_derivatives(d1=0.5)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert model.bss_derivatives(d1=False, d2=False) is None


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['bern', 'spl'])
def test_bss_derivatives_on_gpu(tag):
    check_all_forms(model_from_fixture(tag), tag, 1e-11)


@pytest.mark.gpu
def test_derivative_columns_against_oracle(device_ctx):
    """fokl_build_terms_deriv column by column: splines bit-identical, Bernoulli to the monomial-ulp bound."""
    rng = np.random.default_rng(5)
    n = 3001
    x = rng.random((n, 3))
    x[:4] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1 / 499, 2 / 499, 3 / 499]]
    terms = np.array([[1, 0, 0], [2, 1, 0], [3, 0, 2], [5, 4, 1], [1, 1, 1]], dtype=np.int32)
    for tag in ('spl', 'bern'):
        kname, kid, phis = KINDS[tag]
        packed, nb, width = getKernels.pack_phis(phis, kid)
        device_ctx.upload(x, np.zeros(n), kid, packed, nb, width)
        device_ctx.reserve_slots(2 + len(terms))
        for order, div in ((1, 0.37), (2, 1.9)):
            slots = np.arange(2, 2 + len(terms), dtype=np.int32)
            device_ctx.build_terms_deriv(terms, slots, 0, order, div)
            ob = OracleBackend()
            ob.upload(x[:300], np.zeros(300), kid, packed, nb, width)
            ob.reserve_slots(16)
            ob.build_terms_deriv(terms, list(slots), 0, order, div)
            for j, s in enumerate(slots):
                got = device_ctx.read_slot(int(s))[:300]
                want = ob.cols[int(s)]
                if tag == 'spl':
                    assert np.array_equal(got, want), (order, j)
                else:
                    assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.max(np.abs(want))), (order, j)
