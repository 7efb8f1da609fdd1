"""Pins the oracle (oracle/fokl_oracle.py + oracle_c.c) to the REAL reference through the committed fixtures.

The fixtures under tests/golden/ were produced by importing /root/reference/src in the build container
(tests/golden/make_golden.py).  The oracle restates the reference operation for operation: on the host
that made a fixture (same BLAS / LAPACK / libm behaviour, ``helpers.host_fingerprint``, stored in the .npz) it must
return the reference's numbers to the last bits, for the untouched reference and for the sign-canonical one; on any
other host the sign-canonical fixtures hold to SURVEY 8(c)'s tolerances and the untouched ones as far as no decision
depended on an eigenvector sign.  Every case runs by default (about a minute).
"""
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_case, same_host_as
from oracle import fokl_oracle as O
from fokl_gpy_amd import getKernels

UNITS = np.load(os.path.join(GOLDEN, 'units.npz'))
SPLINE_TAB = np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table']


# ---------------------------------------------------------------------------------------------------------
# F1 / F2 / F3: per-element functions
# ---------------------------------------------------------------------------------------------------------

def test_bernoulli_basis_matches_reference_evaluate_basis():
    bern = getKernels.bernoulli()
    xs = UNITS['bern_x']
    want = UNITS['bern_vals']
    got_scalar = np.array([[O.evaluate_basis(c, np.float64(x), O.KERNEL_BERNOULLI) for x in xs] for c in bern])
    assert np.array_equal(got_scalar, want)
    # C path: one term per order on a single input
    terms = np.arange(1, len(bern) + 1)[:, None]
    got_c = O.build_columns_c(xs[:, None], None, bern, O.KERNEL_BERNOULLI, terms)
    assert np.array_equal(got_c.T, want)


def test_spline_indexing_and_basis_match_reference():
    phis = getKernels.table_to_phis(SPLINE_TAB)
    x = UNITS['spl_x']
    phind, xsm = O.inputs_to_phind(x, len(phis[0][0]))
    assert np.array_equal(phind, UNITS['spl_phind'])
    assert np.array_equal(xsm, UNITS['spl_xsm'])
    terms = np.arange(1, len(phis) + 1)[:, None]
    got_c = O.build_columns_c(xsm, phind, phis, O.KERNEL_SPLINES, terms)
    assert np.array_equal(got_c.T, UNITS['spl_vals'])
    sub = slice(0, 40)
    got_py = O.build_columns_scalar(xsm[sub], phind[sub], phis, O.KERNEL_SPLINES, terms[:5])
    assert np.array_equal(got_py, UNITS['spl_vals'][:5, sub].T)


def test_spline_piece_edges():
    """x = 0 lands on piece 0 (the 0 -> 1 -> 0 quirk), x = 1 on piece 498, knots belong to the piece on their left."""
    x = np.array([[0.0], [1.0], [1 / 499], [2 / 499], [1e-300]])
    phind, xsm = O.inputs_to_phind(x, 499)
    assert phind[:, 0].tolist() == [0, 498, 0, 1, 0]
    assert xsm[0, 0] == 0.0 and xsm[1, 0] == 1.0
    with pytest.raises(ValueError):
        O.inputs_to_phind(np.array([[1.01]]), 499)


def test_scalar_and_c_column_builders_agree_bitwise():
    rng = np.random.default_rng(0)
    x = rng.random((60, 3))
    bern = getKernels.bernoulli()
    terms = np.array([[1, 0, 0], [0, 2, 3], [4, 1, 1], [20, 0, 7]])
    assert np.array_equal(O.build_columns_scalar(x, None, bern, O.KERNEL_BERNOULLI, terms),
                          O.build_columns_c(x, None, bern, O.KERNEL_BERNOULLI, terms))


# ---------------------------------------------------------------------------------------------------------
# F4: enumeration
# ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('key', [k for k in UNITS.files if k.startswith('enum_')])
def test_term_enumeration_matches_np_unique_perms(key):
    pattern = [int(v) for v in key[len('enum_'):].split('_')]
    assert np.array_equal(O.distinct_arrangements(pattern), UNITS[key])


def test_enumeration_beyond_the_reference_limit():
    """M = 16 (configs[3]): counts follow the multinomial formula; the reference's M! enumeration cannot run."""
    from math import comb
    assert O.distinct_arrangements([1] + [0] * 15).shape[0] == 16
    assert O.distinct_arrangements([1, 1] + [0] * 14).shape[0] == comb(16, 2)
    assert O.distinct_arrangements([2, 1] + [0] * 14).shape[0] == 16 * 15
    assert O.distinct_arrangements([1, 1, 1] + [0] * 13).shape[0] == comb(16, 3)
    assert O.distinct_arrangements([2, 1, 1] + [0] * 13).shape[0] == 16 * comb(15, 2)
    rows = O.distinct_arrangements([3, 2, 1] + [0] * 13)
    assert rows.shape[0] == 16 * 15 * 14
    assert np.all(np.diff(np.lexsort(rows.T[::-1])) == 1)           # already in ascending lexicographic order


def test_indvec_progression():
    assert O.deal_indvec(5, 8, 2).tolist() == [3, 2, 0, 0, 0, 0, 0, 0]
    assert O.deal_indvec(7, 6, 3).tolist() == [3, 2, 2, 0, 0, 0]
    assert O.deal_indvec(4, 1, 1).tolist() == [4]


# ---------------------------------------------------------------------------------------------------------
# full fits: oracle == reference (both the untouched reference and the sign-canonical variant)
# ---------------------------------------------------------------------------------------------------------

CASES = ['bern_m1', 'bern_m3_gimmie_tol1', 'bern_m3', 'bern_m4_way3', 'bern_m8_capped', 'bern_m6', 'testdata10_default',
         'testdata10_changed', 'splines_m4', 'sigmoid_splines']
# The reference's 10-row dataset saturates (columns >= rows) after 7 sub-stages: beyond that its numbers are the
# rounding noise of its BLAS; parity is asserted up to that point on any other host.
SATURATES_AFTER = {'testdata10_default': 7, 'testdata10_changed': 7}


def scaled_gram_error(got, want):
    """Largest element difference of two Gram matrices in units of sqrt(diag_i * diag_j): an element that is small by
    cancellation carries the rounding of its summands, which is what a different BLAS blocking changes."""
    d = np.sqrt(np.abs(np.diag(want)))
    return np.max(np.abs(got - want) / np.outer(d, d))


def first_parting(g):
    """Index of the first sub-stage where the fixture's untouched and sign-canonical reference runs part ways (a kill
    test decided differently because an eigenvector came out with the other sign); len(evs) if they never do."""
    r, c = g['ref_evs'], g['canon_evs']
    k = min(len(r), len(c))
    apart = np.nonzero(np.abs(r[:k] - c[:k]) > 1e-9 * np.abs(c[:k]))[0]
    return int(apart[0]) if len(apart) else k


@pytest.mark.parametrize('name', CASES)
def test_oracle_fit_reproduces_the_sign_canonical_reference(name):
    """SURVEY 8(c)'s tolerances, valid on any host: model and call sequence exact, BIC 1e-9 relative, draws 1e-9 of the
    column scale (1e-6 on the sigmoid grid, whose design has a degenerate spectrum), numpy's stream exact, Gram matrices
    1e-13 of sqrt(diag x diag).  On the host that made the fixture (same BLAS / LAPACK / libm behaviour,
    helpers.host_fingerprint) the oracle must ALSO return the reference's numbers to the last bits."""
    g, hy, kname, kid, phis = load_case(name)
    trace = []
    np.random.seed(int(g['seed']))
    betas, mtx, evs = O.fit(g['canon_norm_inputs'], g['canon_norm_data'], phis, kid, eigh=O.eigh_canonical, trace=trace,
                            **hy)
    st = np.random.get_state()
    strict = same_host_as(g)
    upto = len(g['canon_evs']) if strict else SATURATES_AFTER.get(name, len(g['canon_evs']))
    assert np.max(np.abs(evs[:upto] - g['canon_evs'][:upto]) / np.abs(g['canon_evs'][:upto])) <= 1e-9
    if upto < len(g['canon_evs']):
        return                                            # saturated design: nothing beyond this point is defined
    assert mtx.shape == g['canon_mtx'].shape and np.array_equal(mtx, g['canon_mtx'])
    assert len(evs) == len(g['canon_evs'])
    assert [t['cols'] for t in trace] == g['canon_gibbs_sizes'].tolist()
    gb = g['canon_betas']
    err = np.max(np.abs(betas - gb) / np.max(np.abs(gb), axis=0))
    assert err <= (1e-6 if name == 'sigmoid_splines' else 1e-9)
    assert st[2] == int(g['canon_rng_after_fit'][1]) and st[3] == int(g['canon_rng_after_fit'][2])
    for i in range(int(g['canon_n_xtx'])):
        assert scaled_gram_error(trace[i]['xtx'], g[f'canon_xtx_{i}']) <= 1e-13
    if strict:
        assert np.allclose(evs, g['canon_evs'], rtol=1e-13, atol=0)
        assert np.max(np.abs(betas - gb)) <= 1e-12 * np.max(np.abs(gb))
        for i in range(int(g['canon_n_xtx'])):
            assert np.array_equal(trace[i]['xtx'], g[f'canon_xtx_{i}'])


@pytest.mark.parametrize('name', CASES)
def test_oracle_fit_reproduces_the_untouched_reference(name):
    """The untouched reference's kill tests hinge on LAPACK's eigenvector signs, i.e. on the host's library build: its
    numbers are reproducible on the host that made the fixture and nowhere else (the real reference, re-run elsewhere,
    leaves its own fixture at the same sub-stage the oracle does).  Same host: everything, to the last bits.  Another
    host: the BIC trace up to the first sub-stage where the fixture's untouched and sign-canonical runs part -- up to
    there no decision depended on a sign -- then a skip that says so."""
    g, hy, kname, kid, phis = load_case(name)
    trace = []
    np.random.seed(int(g['seed']))
    betas, mtx, evs = O.fit(g['ref_norm_inputs'], g['ref_norm_data'], phis, kid, eigh=O.eigh_reference, trace=trace, **hy)
    if same_host_as(g):
        assert mtx.shape == g['ref_mtx'].shape and np.array_equal(mtx, g['ref_mtx'])
        assert np.allclose(evs, g['ref_evs'], rtol=1e-13, atol=0)
        assert np.max(np.abs(betas - g['ref_betas'])) <= 1e-12 * np.max(np.abs(g['ref_betas']))
        assert [t['cols'] for t in trace] == g['ref_gibbs_sizes'].tolist()
        for i in range(int(g['ref_n_xtx'])):
            assert np.array_equal(trace[i]['xtx'], g[f'ref_xtx_{i}'])
        return
    k = min(first_parting(g), SATURATES_AFTER.get(name, 1 << 30), len(evs))
    assert k >= 1 and np.max(np.abs(evs[:k] - g['ref_evs'][:k]) / np.abs(g['ref_evs'][:k])) <= 1e-9
    made_on = str(g['host_fingerprint']) if 'host_fingerprint' in g.files else 'unrecorded'
    pytest.skip(f"fixture made on a host with another BLAS / LAPACK build ({made_on}): BIC trace equal "
                f"over the {k} sign-independent sub-stages, the rest of the untouched reference is host-specific")


def test_reference_itself_depends_on_eigenvector_signs():
    """Documented caveat: the untouched reference and its sign-canonical variant may select different
    models because kill tests hinge on Monte-Carlo statistics.  Parity is therefore pinned on the canonical variant."""
    g, *_ = load_case('bern_m3')
    assert not (g['ref_mtx'].shape == g['canon_mtx'].shape and np.array_equal(g['ref_mtx'], g['canon_mtx']))
    g1, *_ = load_case('bern_m1')
    assert np.array_equal(g1['ref_mtx'], g1['canon_mtx'])


def test_evaluate_and_coverage_match_reference():
    g, hy, kname, kid, phis = load_case('bern_m3')
    mean, bounds = O.evaluate(g['canon_norm_inputs'], g['canon_betas'], g['canon_mtx'], phis, kid, hy['draws'],
                              g['canon_setnos'], return_bounds=True)
    assert np.allclose(mean, g['canon_cov_mean'], rtol=1e-12, atol=1e-14)
    assert np.allclose(bounds, g['canon_cov_bounds'], rtol=1e-12, atol=1e-14)
    assert abs(O.coverage_rmse(mean, g['canon_norm_data']) - float(g['canon_cov_rmse'])) < 1e-12
