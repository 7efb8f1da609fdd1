"""GP_Integrate, the consumer of fitted models -- against a trajectory computed by the reference's own
GP_Integrate (tests/golden/make_golden.py gp_integrate; host code, no GPU needed)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from fokl_gpy_amd import getKernels
from fokl_gpy_amd.GP_Integrate import GP_Integrate


def _case():
    g = np.load(os.path.join(GOLDEN, 'gp_integrate.npz'))
    phis = getKernels.table_to_phis(np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table'])
    betas = [g['betas0'], g['betas1']]
    mtx = [g['mtx0'], g['mtx1']]
    return g, phis, betas, mtx


def test_trajectory_matches_the_reference():
    g, phis, betas, mtx = _case()
    y0 = g['y0'].copy()
    T, Y = GP_Integrate(betas, mtx, g['b'], g['norms'], phis, float(g['start']), float(g['stop']), y0, float(g['h']),
                        [row for row in g['used']])
    assert np.array_equal(T, g['T']) and Y.shape == g['Y'].shape
    assert np.array_equal(Y[:, 0], g['y0'])
    np.testing.assert_allclose(Y, g['Y'], rtol=0, atol=1e-13)
    assert np.array_equal(y0, Y[:, -1])                       # y0 is advanced in place, as in the reference
    np.testing.assert_allclose(y0, g['y_after'], rtol=0, atol=1e-13)
    # the state saturates at the upper bound of norms for a while in this case: the clamps are exercised
    assert (g['Y'][0] >= g['norms'][1, 0]).any()


def test_models_that_ignore_the_forcing_and_single_state():
    g, phis, betas, mtx = _case()
    # one state, no forcing: dy/dt = model(y) with a 1-column matrix
    m1 = np.array([[1.0], [2.0]])
    b1 = np.array([0.2, -0.4, 0.1])
    y0 = np.array([0.4])
    T, Y = GP_Integrate([b1], [m1], np.zeros((0,)), np.array([[0.0], [1.0]]), phis, 0.0, 1.0, y0, 0.1, [np.array([1])])
    assert Y.shape == (1, len(T)) and np.isfinite(Y).all()
    # plain RK4 of the same right-hand side in numpy
    tab = np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table']

    def basis(order, x):
        p = min(int(np.floor(x * 498)), 497)
        X = (x - p / 498) / (1 / 498)
        c = tab[order - 1, :, p]
        return c[0] + c[1] * X + c[2] * X ** 2 + c[3] * X ** 3

    def f(y):
        x = min(max(y, 0.0), 1.0)
        return b1[0] + b1[1] * basis(1, x) + b1[2] * basis(2, x)

    y, want = 0.4, [0.4]
    for _ in range(len(T) - 1):
        k1 = 0.1 * f(y); k2 = 0.1 * f(y + k1 / 2); k3 = 0.1 * f(y + k2 / 2); k4 = 0.1 * f(y + k3)
        y += (k1 + 2 * k2 + 2 * k3 + k4) / 6
        want.append(y)
    np.testing.assert_allclose(Y[0], want, rtol=1e-12, atol=1e-14)


def test_argument_errors():
    g, phis, betas, mtx = _case()
    args = (g['norms'], phis, 2.0, 3.0)
    with pytest.raises(IndexError):            # re-ordering entries: broken in the reference, refused here
        GP_Integrate(betas, mtx, g['b'], *args, g['y0'].copy(), 0.05, [np.array([2, 1, 3]), np.array([1, 1, 1])])
    with pytest.raises(IndexError):            # forcing shorter than the integration
        GP_Integrate(betas, mtx, g['b'][:3], *args, g['y0'].copy(), 0.05, [row for row in g['used']])
    with pytest.raises(IndexError):            # a model with three input columns that is routed two inputs
        GP_Integrate(betas, mtx, g['b'], *args, g['y0'].copy(), 0.05, [np.array([1, 1, 0]), np.array([1, 1, 1])])
    with pytest.raises(ValueError):
        GP_Integrate(betas[:1], mtx, g['b'], *args, g['y0'].copy(), 0.05, [row for row in g['used']])
