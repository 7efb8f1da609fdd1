"""G2 of a kill test's model derived from the eigenpairs of the model it is tested against (fokl_pool_submit_spectral_update,
csrc/fokl_hostpool.cpp: secular equation + one product) along a CHAIN of deletions -- each model derived from the result
before it, as the accepted tests of a sub-stage are, forty deep (the search cuts the chain at 24) -- against a fresh
decomposition of every model; and the rule that hands numerically singular models to the reference's own driver
(FOKL_EIGH_SINGULAR).  What tests/stress/eigen_update_chain.py prints by hand, asserted."""
import os
import sys

import numpy as np
import pytest
import scipy.linalg

from helpers import ROOT
from fokl_gpy_amd import _capi, engine

sys.path.insert(0, os.path.join(ROOT, 'tests', 'stress'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _gram(kind, n, rng):
    """[n + 1, n + 1] with y last: synthetic columns of mixed scales (condition number ~1e4), or the Gram of Bernoulli terms
    over the configs[2] inputs at 2e4 rows (the benchmark's own conditioning: 1e7-1e8 at 100 columns)."""
    if kind == 'synthetic':
        from eigh_device_probe import gram_like
        return gram_like(n + 1, rng)
    from eigen_deletion_study import real_gram
    A = real_gram(n, rng, rows=20_000)
    g = np.zeros((n + 1, n + 1))
    g[:n, :n] = A
    g[:n, n] = g[n, :n] = rng.standard_normal(n) * 100
    g[n, n] = 1e6
    return g


@pytest.mark.parametrize('kind,n', [('synthetic', 90), ('bernoulli', 100)])
def test_chain_of_derived_eigenpairs_stays_at_the_fresh_decompositions(kind, n):
    rng = np.random.default_rng(5)
    gram = _gram(kind, n, rng)
    cond = np.linalg.cond(gram[:n, :n])
    eps = np.finfo(np.float64).eps
    pool = _capi.HostPool(_capi.LegacyStream(np.random.RandomState(1).get_state()), chain_threads=1, spectral_threads=1)
    try:
        assert pool.has_dgemm
        alive = np.arange(n, dtype=np.int32)
        parent = pool.submit_spectral(gram, alive, n).wait()
        derived = 0
        worst = dict(noise=0.0, lamb=0.0, orth=0.0, beta=0.0, moments=0.0)
        for step in range(40):
            c = int(rng.integers(1, alive.shape[0]))
            child = np.ascontiguousarray(np.delete(alive, c))
            job, updated = pool.submit_spectral_update(gram, child, n, parent, c)
            res = job.wait()
            fresh = pool.submit_spectral(gram, child, n).wait()
            m = child.shape[0]
            derived += int(updated[0])
            # the chain's noise map Q diag((lam + 1 / tau^2)^-1/2) at tau^2 = 1: what a draw is made of (FR:1525-1528)
            M0, M1 = fresh.Qt.T / np.sqrt(fresh.lamb + 1), res.Qt.T / np.sqrt(res.lamb + 1)
            worst['noise'] = max(worst['noise'], np.abs(M0 - M1).max() / np.abs(M0).max())
            worst['lamb'] = max(worst['lamb'], np.abs(res.lamb - fresh.lamb).max() / fresh.lamb.max())
            worst['orth'] = max(worst['orth'], np.abs(res.Qt @ res.Qt.T - np.eye(m)).max())
            worst['beta'] = max(worst['beta'], np.abs(res.betahat - fresh.betahat).max() / np.abs(fresh.betahat).max())
            worst['moments'] = max(worst['moments'], abs(res.moments[1] - fresh.moments[1]) / abs(fresh.moments[1]))
            parent, alive = res, child
        assert derived >= 36, f"only {derived} of 40 models were derived (the rest decomposed afresh)"
        assert worst['orth'] <= 1e-13                       # orthogonal to working precision whatever the conditioning
        assert worst['lamb'] <= 64 * eps                    # eigenvalues: relative to the largest
        # eigenvectors turn by eps ||A|| / gap under a rounding-level change of A -- two decompositions of one matrix are that
        # far apart too; no growth along the chain: the bound is the one-step bound
        assert worst['noise'] <= 64 * eps * cond, (worst, cond)
        assert worst['beta'] <= 64 * eps * cond
        assert worst['moments'] <= 1e-9                     # the BIC the eigenpairs bring along (held against the decision's)
    finally:
        pool.close()


def _matrix_with_spectrum(lamb, rng):
    q, _ = np.linalg.qr(rng.standard_normal((lamb.shape[0], lamb.shape[0])))
    a = (q * lamb) @ q.T
    return 0.5 * (a + a.T)


@pytest.mark.parametrize('ratio,reference_driver', [(4e-9, False), (2.5e-10, True)])
def test_numerically_singular_models_get_the_reference_driver(ratio, reference_driver):
    """Smallest eigenvalue <= 1e-9 of the largest (FOKL_EIGH_SINGULAR): the eigenvectors of the (near) null space are the
    driver's choice, so the model is decomposed by dsyevr, as scipy.linalg.eigh does for the reference (FR:1499), never by
    dsyevd nor derived from its parent; just above the threshold both shortcuts apply.  Either side of the edge."""
    rng = np.random.default_rng(11)
    n = 96                                                  # (>= 80 columns: dsyevd territory, FOKL_EIGH_DC_FROM)
    lamb = np.concatenate([[ratio], np.geomspace(1e-3, 1.0, n - 1)]) * 1e6
    a = _matrix_with_spectrum(lamb, rng)
    gram = np.zeros((n + 1, n + 1))
    gram[:n, :n] = a
    gram[:n, n] = gram[n, :n] = rng.standard_normal(n)
    gram[n, n] = 1e3
    pool = _capi.HostPool(_capi.LegacyStream(np.random.RandomState(1).get_state()), chain_threads=1, spectral_threads=1)
    try:
        idx = np.arange(n, dtype=np.int32)
        res = pool.submit_spectral(gram, idx, n).wait()
        w, v = scipy.linalg.eigh(gram[:n, :n], driver='evr')       # the reference's call
        v = v * np.where(v[np.argmax(np.abs(v), axis=0), np.arange(n)] < 0, -1.0, 1.0)
        same_as_reference = np.array_equal(res.lamb, w) and np.array_equal(res.Qt, v.T)
        assert same_as_reference == reference_driver or not reference_driver and same_as_reference is False
        if reference_driver:
            assert same_as_reference
        # and a model derived from it: refused below the edge (decomposed afresh by dsyevr), taken above
        child = np.ascontiguousarray(np.delete(idx, 5))
        # (deleting a column leaves the ratio about where it was: interlacing)
        job, updated = pool.submit_spectral_update(gram, child, n, res, 5)
        job.wait()
        assert int(updated[0]) == (0 if reference_driver else 1)
    finally:
        pool.close()
