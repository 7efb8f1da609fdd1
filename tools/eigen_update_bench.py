"""tools/eigen_update_bench.py: one spectral thread, one model per size -- G2 by fokl_pool_submit_spectral_update (from the
eigenpairs of the model with one more column: secular equation + one dgemm) against a fresh decomposition (dsyevr below
FOKL_EIGH_DC_FROM columns, dsyevd from there on) of the same model; best of 25 round trips each, so the ~30 us of the
Python call and the thread's wake-up are in both; BLAS on one thread, as inside a fit (FoKLRoutines caps it).  Also the
deviation between the two in the chain's noise map."""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np

from fokl_gpy_amd import _capi
from eigh_device_probe import gram_like


def best(f, reps=25):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        b = min(b, time.perf_counter() - t0)
    return 1e3 * b


def main():
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1, user_api='blas'):
        measure()


def measure():
    rng = np.random.default_rng(5)
    np.random.seed(1)
    # EUB_DEVICE_DGEMM=columns: the update's product on the device (fokl_device_dgemm) from that many columns on
    dev = os.environ.get('EUB_DEVICE_DGEMM')
    pool = _capi.HostPool(_capi.LegacyStream(), chain_threads=1, spectral_threads=1,
                          device_dgemm=(0, int(dev)) if dev else None)
    print(f"dgemm bound: {pool.has_dgemm} (on the device from {getattr(pool, 'device_dgemm_from', 0)} columns; 0: never); "
          f"dsyevd from {pool.dsyevd_from} columns")
    print(f"{'columns':>8s} {'update ms':>10s} {'fresh ms':>9s} {'ratio':>6s} {'map deviation':>14s} {'orthogonality':>14s}")
    for n in [int(v) for v in os.environ.get('EUB_SIZES', '24,48,66,80,96,112,128,144').split(',')]:
        gram = gram_like(n + 1, rng)
        idx = np.arange(n, dtype=np.int32)
        par = pool.submit_spectral(gram, idx, n).wait()
        c = n // 2
        child = np.ascontiguousarray(np.delete(idx, c))
        tu = best(lambda: pool.submit_spectral_update(gram, child, n, par, c)[0].wait())
        tf = best(lambda: pool.submit_spectral(gram, child, n).wait())
        job, updated = pool.submit_spectral_update(gram, child, n, par, c)
        u = job.wait()
        f = pool.submit_spectral(gram, child, n).wait()
        assert updated[0] == 1
        Mu, Mf = u.Qt.T / np.sqrt(u.lamb + 1.0), f.Qt.T / np.sqrt(f.lamb + 1.0)
        print(f"{n - 1:8d} {tu:10.3f} {tf:9.3f} {tf / tu:6.1f} {np.abs(Mu - Mf).max() / np.abs(Mf).max():14.2e} "
              f"{np.abs(u.Qt @ u.Qt.T - np.eye(n - 1)).max():14.2e}")
    pool.close()


if __name__ == '__main__':
    main()
