"""Feasibility study (numpy prototype, CPU only): the eigen-decomposition of a kill test's model from its PARENT's instead of
from scratch.

A kill test's XtX is its parent's with row and column c deleted.  With XtX = Q diag(lam) Q' and z = row c of Q:
  * the eigenvalues mu_k of the deleted matrix are the n - 1 roots of g(mu) = sum_j z_j^2 / (lam_j - mu), one in each
    interval (lam_k, lam_k+1);
  * its eigenvectors are the rows != c of Q x_k, x_k[j] = z_j / (lam_j - mu_k) normalised -- one (n-1) x n x (n-1) product;
  * computed roots are exact roots for a slightly different z (Gu & Eisenstat): z^_j^2 = prod_k (mu_k - lam_j) /
    prod_{i != j} (lam_i - lam_j), which makes the x_k orthogonal to working precision.
O(n^2) + one matrix product: 0.15 ms at 140 columns on a host thread against dsyevd's 0.8 ms, tens of microseconds on the
device.  The question this script answers: how far is the result from a fresh decomposition after a CHAIN of deletions (a
sub-stage's kill tests delete up to ~30 columns one after the other)?  Measured on Gram matrices with the conditioning of
the benchmark fit's models; reported as the deviation of the chain's noise map Q diag((lam + 1)^-1/2) from a fresh LAPACK
decomposition of the same sub-block and from an 80-bit Jacobi reference, relative to the map's largest entry.

    python tests/stress/eigen_deletion_study.py [columns] [deletions] [decades of column scale | real]

'real': XtX of Bernoulli-kernel terms (the oracle's column builder) on the configs[2] dataset at 1e5 rows -- the final
model of the committed golden widened with low-order terms to the requested width, as a wide sub-stage's models are.
"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import scipy.linalg as sl

from eigh_device_probe import gram_like, jacobi_truth, canonical, draw_map

EPS = np.finfo(float).eps


def secular_roots(lam, z2):
    """Roots of sum_j z2_j / (lam_j - mu) in (lam_k, lam_k+1), k = 0 .. n - 2, as (origin index, tau): mu = lam[origin] + tau
    with the origin the nearer pole, so that lam_j - mu = (lam_j - lam[origin]) - tau is formed without cancellation."""
    n = lam.shape[0]
    origin = np.empty(n - 1, dtype=np.int64)
    tau = np.empty(n - 1)
    for k in range(n - 1):
        gap = lam[k + 1] - lam[k]
        mid = lam[k] + 0.5 * gap
        gmid = np.sum(z2 / (lam - mid))
        o = k if gmid > 0.0 else k + 1                       # g rises from -inf to +inf across the interval
        d0 = lam - lam[o]
        lo, hi = (0.0, 0.5 * gap) if o == k else (-0.5 * gap, 0.0)
        t = 0.5 * (lo + hi)
        for _ in range(200):
            delta = d0 - t
            terms = z2 / delta
            g = np.sum(terms)
            if abs(g) <= 4 * n * EPS * np.sum(np.abs(terms)):
                break
            if g > 0.0:
                hi = t
            else:
                lo = t
            # the two nearest poles exactly, the rest frozen (the "middle way" of the divide-and-conquer literature)
            rest = g - terms[k] - terms[k + 1]
            a, b = z2[k], z2[k + 1]
            dk, dk1 = d0[k], d0[k + 1]
            # a / (dk - t) + b / (dk1 - t) + rest = 0  ->  rest (dk - t)(dk1 - t) + a (dk1 - t) + b (dk - t) = 0
            qa, qb, qc = rest, -(rest * (dk + dk1) + a + b), rest * dk * dk1 + a * dk1 + b * dk
            cand = None
            if qa == 0.0:
                if qb != 0.0:
                    cand = -qc / qb
            else:
                disc = qb * qb - 4 * qa * qc
                if disc >= 0.0:
                    sq = np.sqrt(disc)
                    r1 = (-qb - np.sign(qb if qb != 0 else 1.0) * sq) / (2 * qa)
                    r2 = qc / (qa * r1) if r1 != 0.0 else np.inf
                    cand = r1 if lo < r1 < hi else (r2 if lo < r2 < hi else None)
            t_new = cand if cand is not None and lo < cand < hi else 0.5 * (lo + hi)
            if t_new == t:
                break
            t = t_new
        origin[k], tau[k] = o, t
    return origin, tau


def delete_column(lam, Q, c):
    """(lam, Q) of A -> (lam', Q') of A without row / column c.  No deflation beyond tiny z (the study's matrices have none
    of the close eigenvalue pairs a production version must rotate away)."""
    n = lam.shape[0]
    z = Q[c, :].copy()
    keep = np.abs(z) > 64 * EPS                                  # z_j = 0: lam_j stays an eigenvalue, q_j (row c dropped) its vector
    rows = np.arange(n) != c
    if not np.all(keep):
        raise NotImplementedError('deflation is not part of the prototype')
    z2 = z * z
    origin, tau = secular_roots(lam, z2)
    # delta[j, k] = lam_j - mu_k without cancellation
    delta = (lam[:, None] - lam[origin][None, :]) - tau[None, :]
    # Gu-Eisenstat: the z for which the computed roots are exact (products of ratios in (0, ...): interlacing)
    zh2 = np.empty(n)
    for j in range(n):
        num = -delta[j, :]                                        # mu_k - lam_j
        den = np.delete(lam, j) - lam[j]                          # lam_i - lam_j, i != j
        zh2[j] = np.prod(num / den)
    zh = np.sqrt(np.abs(zh2)) * np.sign(z)
    X = zh[:, None] / delta                                       # x_k[j] = z^_j / (lam_j - mu_k)
    X /= np.sqrt(np.sum(X * X, axis=0))[None, :]
    mu = lam[origin] + tau
    return mu, Q[rows, :] @ X


def real_gram(columns, rng, rows=100_000):
    import bench
    from fokl_gpy_amd import getKernels
    from oracle import fokl_oracle as O
    x, _, _ = bench.config_workload(2, 0, rows)
    x = (x - x.min(axis=0)) / (x.max(axis=0) - x.min(axis=0))
    mtx = np.load(os.path.join(ROOT, 'tests', 'golden', 'cfg2_n1e6_m8.npz'))['mtx'].astype(np.int32)
    have = {tuple(r) for r in mtx}
    extra = []
    while len(have) + 1 < columns:
        r = np.zeros(x.shape[1], dtype=np.int32)
        for i in rng.choice(x.shape[1], size=int(rng.integers(1, 3)), replace=False):
            r[i] = int(rng.integers(1, 5))
        if tuple(r) not in have:
            have.add(tuple(r))
            extra.append(r)
    terms = np.vstack([mtx] + extra)[:columns - 1]
    X = O.build_columns_c(x, None, getKernels.bernoulli(), O.KERNEL_BERNOULLI, terms, threads=8)
    X = np.column_stack([np.ones(rows), X])
    return X.T @ X


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    deletions = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rng = np.random.default_rng(5)
    if len(sys.argv) > 3 and sys.argv[3] == 'real':
        A = real_gram(n, rng)
    else:
        spread = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
        A = gram_like(n + 1, rng, spread=spread)[:n, :n]
    print(f"{n} columns, condition number {np.linalg.cond(A):.2e}; deviation of the noise map from a fresh decomposition of the "
          f"same sub-block, relative to its largest entry")
    print(f"{'deleted':>8s} {'columns':>8s} {'chain vs dsyevr':>16s} {'chain vs truth':>15s} {'dsyevr vs truth':>16s} "
          f"{'orthogonality':>14s} {'residual':>10s} {'ms (numpy)':>11s}")
    lam, Q = sl.eigh(A)
    alive = np.arange(n)
    for step in range(1, deletions + 1):
        c = int(rng.integers(1, alive.shape[0]))                  # never the intercept
        t0 = time.perf_counter()
        lam, Q = delete_column(lam, Q, c)
        ms = 1e3 * (time.perf_counter() - t0)
        alive = np.delete(alive, c)
        sub = A[np.ix_(alive, alive)]
        if step in (1, 2, 5, 10, 20, 30, 40, 60) or step == deletions:
            fl, fQ = sl.eigh(sub)
            tl, tQ = jacobi_truth(sub)
            T = draw_map(tl, canonical(tQ))
            scale = np.abs(T).max()
            mine = draw_map(lam, canonical(Q))
            fresh = draw_map(fl, canonical(fQ))
            m = alive.shape[0]
            print(f"{step:8d} {m:8d} {np.abs(mine - fresh).max() / scale:16.2e} {np.abs(mine - T).max() / scale:15.2e} "
                  f"{np.abs(fresh - T).max() / scale:16.2e} {np.abs(Q.T @ Q - np.eye(m)).max():14.2e} "
                  f"{np.abs(sub @ Q - Q * lam).max() / np.abs(sub).max():10.2e} {ms:11.2f}", flush=True)


if __name__ == '__main__':
    main()
