"""G2 on the device against LAPACK: time, sweeps and accuracy of spectral_jacobi_kernel per model size (GPU box).

    python tools/eigh_device_probe.py > profiles/eigh_device_rNN.txt

Accuracy is measured against a Jacobi iteration in 80-bit arithmetic on the host (the "truth" column): max deviation of
the map a chain applies to its noise, Q diag((lamb + 1)^-1/2), relative to its largest entry -- for LAPACK's dsyevr (the
reference's solver, FR:1499), for the device kernel, and for dsyevr on a Gram whose entries moved by one unit in the
last place (what any other BLAS build does to the reference itself).
"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.linalg as sl

from fokl_gpy_amd import _capi


def gram_like(cols, rng, rows=4000, spread=2.0):
    X = rng.standard_normal((rows, cols - 1)) * 10.0 ** rng.uniform(-spread / 2, spread / 2, cols - 1)
    X[:, ::3] += 0.7 * X[:, :1]
    y = X @ rng.standard_normal(cols - 1) * 0.1 + rng.standard_normal(rows)
    Z = np.column_stack([np.ones(rows), X, y])
    return Z.T @ Z


def jacobi_truth(A):
    L = np.longdouble
    A = A.astype(L).copy()
    n = A.shape[0]
    V = np.eye(n, dtype=L)
    eps = np.finfo(L).eps
    for sweep in range(40):
        rot = 0
        for p in range(n - 1):
            for q in range(p + 1, n):
                apq = A[p, q]
                if abs(apq) <= eps * np.sqrt(abs(A[p, p] * A[q, q])):
                    continue
                rot += 1
                theta = (A[q, q] - A[p, p]) / (2 * apq)
                t = (1 if theta >= 0 else -1) / (abs(theta) + np.sqrt(theta * theta + 1))
                c = 1 / np.sqrt(t * t + 1)
                s = t * c
                for M in (A, V):
                    Mp, Mq = M[:, p].copy(), M[:, q].copy()
                    M[:, p], M[:, q] = c * Mp - s * Mq, s * Mp + c * Mq
                Ap, Aq = A[p, :].copy(), A[q, :].copy()
                A[p, :], A[q, :] = c * Ap - s * Aq, s * Ap + c * Aq
        if rot == 0:
            break
    lam = np.diag(A).copy()
    o = np.argsort(lam)
    return lam[o].astype(float), V[:, o].astype(float)


def canonical(Q):
    Q = Q.copy()
    for j in range(Q.shape[1]):
        p = np.argmax(np.abs(Q[:, j]))
        if Q[p, j] < 0:
            Q[:, j] = -Q[:, j]
    return Q


def draw_map(lam, Q):
    return Q / np.sqrt(lam + 1.0)


def main():
    eng = _capi.DeviceSpectralEngine(int(os.environ.get('FOKL_DEVICE', '0')))
    rng = np.random.default_rng(3)
    print(f"{'n':>4s} {'sweeps':>6s} {'rotations':>9s} {'device us':>10s} {'16 at once us':>13s} {'dsyevr us':>10s} "
          f"{'cond':>9s} {'dsyevr-truth':>12s} {'device-truth':>12s} {'dsyevr(ulp)-dsyevr':>18s}")
    for n in (4, 8, 16, 24, 32, 40, 48, 56, 64, 66, 72, 80, 96, 112, 128, 144, 160, 192):
        gram = gram_like(n + 1, rng)
        idx = np.arange(n, dtype=np.int32)
        A = gram[:n, :n]
        for _ in range(2):
            job = eng.submit(gram, idx)
            lamb, Qt, *_ = job.wait()
            info = job.info()
            lamb, Qt = lamb.copy(), Qt.copy()
            job.release()
        jobs = [eng.submit(gram, idx, launch=False) for _ in range(16)]
        t0 = time.perf_counter()
        eng.flush()
        for j in jobs:
            j.wait()
        batch = (time.perf_counter() - t0) * 1e6
        for j in jobs:
            j.release()
        t0 = time.perf_counter()
        for _ in range(20):
            ref_l, ref_Q = sl.eigh(A)
        lapack = (time.perf_counter() - t0) / 20 * 1e6
        ref_Q = canonical(ref_Q)
        tl, tQ = jacobi_truth(A)
        tQ = canonical(tQ)
        u = rng.uniform(-1, 1, A.shape)
        pl, pQ = sl.eigh(A * (1 + 2.0 ** -52 * (u + u.T) / 2))
        pQ = canonical(pQ)
        T = draw_map(tl, tQ)
        scale = np.abs(T).max()
        print(f"{n:4d} {info['sweeps']:6d} {info['rotations']:9d} {info['seconds'] * 1e6:10.1f} {batch:13.1f} {lapack:10.1f} "
              f"{np.linalg.cond(A):9.2e} {np.abs(draw_map(ref_l, ref_Q) - T).max() / scale:12.2e} "
              f"{np.abs(draw_map(lamb, Qt.T) - T).max() / scale:12.2e} "
              f"{np.abs(draw_map(pl, pQ) - draw_map(ref_l, ref_Q)).max() / scale:18.2e}", flush=True)
    eng.close()


if __name__ == '__main__':
    main()
