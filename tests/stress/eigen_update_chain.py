"""The native eigen-update (fokl_pool_submit_spectral_update, csrc/fokl_hostpool.cpp) along a CHAIN of deletions -- each
model derived from the result before, as the kill tests of a sub-stage do -- against a fresh decomposition of every model:
deviation in the chain's noise map Q diag((lam + 1)^-1/2), eigenvalues, orthogonality, betahat, residual moments.  CPU only.

    python tests/stress/eigen_update_chain.py [columns] [real]

'real': the Gram of Bernoulli terms over the configs[2] dataset at 1e5 rows (eigen_deletion_study.real_gram: condition number
1.5e8 at 140 columns), else a synthetic one (tools/eigh_device_probe.gram_like: 1e4).  The output of `140 real` is the last
block of profiles/eigen_update_r04.txt."""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

from fokl_gpy_amd import _capi
from eigen_deletion_study import real_gram
from eigh_device_probe import gram_like


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 140
    rng = np.random.default_rng(5)
    if len(sys.argv) > 2 and sys.argv[2] == 'real':
        A = real_gram(n, rng)
        yv = rng.standard_normal(n) * 100
        gram = np.zeros((n + 1, n + 1))
        gram[:n, :n] = A
        gram[:n, n] = gram[n, :n] = yv
        gram[n, n] = 1e6
    else:
        gram = gram_like(n + 1, rng)
    print(f"{n} columns, condition number {np.linalg.cond(gram[:n, :n]):.2e}")
    np.random.seed(1)
    pool = _capi.HostPool(_capi.LegacyStream(), chain_threads=1, spectral_threads=1)
    try:
        alive = np.arange(n, dtype=np.int32)
        parent = pool.submit_spectral(gram, alive, n).wait()
        print(f"{'step':>5s} {'columns':>8s} {'derived':>8s} {'noise map':>10s} {'eigenvalues':>12s} {'orthogonality':>14s} "
              f"{'betahat':>9s} {'moments':>9s}")
        for step in range(1, 31):
            c = int(rng.integers(1, alive.shape[0]))
            child = np.ascontiguousarray(np.delete(alive, c))
            job, updated = pool.submit_spectral_update(gram, child, n, parent, c)
            res = job.wait()
            fresh = pool.submit_spectral(gram, child, n).wait()
            m = child.shape[0]
            M0, M1 = fresh.Qt.T / np.sqrt(fresh.lamb + 1), res.Qt.T / np.sqrt(res.lamb + 1)
            print(f"{step:5d} {m:8d} {int(updated[0]):8d} {np.abs(M0 - M1).max() / np.abs(M0).max():10.2e} "
                  f"{np.abs(res.lamb - fresh.lamb).max() / fresh.lamb.max():12.2e} "
                  f"{np.abs(res.Qt @ res.Qt.T - np.eye(m)).max():14.2e} "
                  f"{np.abs(res.betahat - fresh.betahat).max() / np.abs(fresh.betahat).max():9.2e} "
                  f"{np.abs(res.moments - fresh.moments).max() / np.abs(fresh.moments).max():9.2e}")
            parent, alive = res, child
    finally:
        pool.close()


if __name__ == '__main__':
    main()
