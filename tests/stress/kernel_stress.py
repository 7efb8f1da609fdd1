"""Random shapes through K1 / K2 / K3 / predict on the device against the oracle's C restatement and numpy (development
aid / stress run; tolerances as in tests/test_gpu_parity.py)."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fokl_gpy_amd import _capi, getKernels
from oracle import fokl_oracle as O
BERN = getKernels.bernoulli()
SPL = getKernels.table_to_phis(np.load(os.path.join(ROOT, 'tests', 'golden', 'spline_phis.npz'))['table'])
ctx = _capi.DeviceContext(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for trial in range(trials):
    kid = int(rng.integers(0, 2))
    phis = SPL if kid == O.KERNEL_SPLINES else BERN
    n = int(rng.choice([1, 2, 63, 64, 65, 511, 513, 1000, 4099, 20011, 100003]))
    m = int(rng.integers(1, 13))
    x = rng.random((n, m))
    if rng.integers(0, 3) == 0:
        x[rng.integers(0, n)] = 0.0; x[rng.integers(0, n)] = 1.0          # the ends of the interval
    y = rng.standard_normal(n)
    packed, nb, width = getKernels.pack_phis(phis, kid)
    ctx.upload(x, y, kid, packed, nb, width)
    T = int(rng.integers(1, 90))
    max_order = len(phis) if kid == O.KERNEL_SPLINES else int(rng.choice([3, 6, 20]))
    terms = np.zeros((T, m), dtype=np.int32)
    for t in range(T):
        k = int(rng.integers(1, min(m, 4) + 1))
        for c in rng.choice(m, size=k, replace=False):
            terms[t, c] = int(rng.integers(1, max_order + 1))
    ctx.reserve_slots(2 + T)
    slots = np.arange(2, 2 + T, dtype=np.int32)
    ctx.build_terms(terms, slots)
    got = np.stack([ctx.read_slot(int(s)) for s in slots], axis=1)
    if kid == O.KERNEL_SPLINES:
        phind, xsm = O.inputs_to_phind(x, len(phis[0][0]))
    else:
        phind, xsm = None, x
    want = O.build_columns_c(xsm, phind, phis, kid, terms)
    if kid == O.KERNEL_SPLINES:
        # bit-identical except where glibc's pow (the reference's t**2, t**3) is off by an ulp: about 1 evaluation in 1e5
        off = got != want
        ok1 = bool(np.all(np.abs(got - want) <= 8 * 2.0 ** -52 * np.abs(want)) and off.mean() < 1e-3)
    else:
        bound = np.ones((n, T))
        for j, term in enumerate(terms):
            for k, o in enumerate(term):
                if o:
                    c = np.abs(np.asarray(BERN[o - 1]))
                    bound[:, j] *= sum(c[p] * np.abs(x[:, k]) ** p for p in range(len(c)))
        ok1 = bool(np.all(np.abs(got - want) <= 2.0 ** -50 * bound * 4))
    # K2: random row / column subsets, against numpy
    cols = np.concatenate([np.ones((n, 1)), got, y[:, None]], axis=1)
    allslots = np.concatenate([[0], slots, [1]]).astype(np.int32)
    nr, nc = int(rng.integers(1, min(T, 70) + 1)), int(rng.integers(1, T + 3))
    ri = rng.choice(T, size=nr, replace=False) + 1
    ci = rng.choice(T + 2, size=nc, replace=False)
    g = ctx.gram(allslots[ri], allslots[ci])
    gw = cols[:, ri].T @ cols[:, ci]
    scale = np.sqrt(np.sum(cols[:, ri] ** 2, axis=0))[:, None] * np.sqrt(np.sum(cols[:, ci] ** 2, axis=0))[None, :] + 1e-300
    ok2 = bool(np.max(np.abs(g - gw) / scale) < 1e-12)
    # K3: residual moments of a random subset
    k3 = rng.choice(T + 1, size=int(rng.integers(1, T + 2)), replace=False)
    beta = rng.standard_normal(len(k3))
    s1, s2 = ctx.bic_resid(allslots[k3], beta)
    r = y - cols[:, k3] @ beta
    ok3 = abs(s1 - r.sum()) <= 1e-11 * (np.abs(r).sum() + 1) and abs(s2 - (r * r).sum()) <= 1e-11 * ((r * r).sum() + 1)
    # predict with bounds
    pc = rng.choice(T + 1, size=int(rng.integers(1, min(T + 1, 60) + 1)), replace=False)
    draws = int(rng.choice([5, 40, 64, 77, 250, 1000]))
    betas = rng.standard_normal((draws, len(pc))) * rng.choice([0.01, 1.0])
    cut = int(np.floor(draws * 0.025) + 1)
    ok4 = True
    if cut < draws:
        mean, bounds = ctx.predict(allslots[pc], betas, cut)
        mod = cols[:, pc] @ betas.T
        srt = np.sort(mod, axis=1)
        sc = np.abs(mod).max(axis=1) + 1e-300
        ok4 = bool(np.max(np.abs(mean - mod.mean(1)) / sc) < 1e-12 and np.max(np.abs(bounds[:, 0] - srt[:, cut]) / sc) < 1e-12
                   and np.max(np.abs(bounds[:, 1] - srt[:, draws - cut]) / sc) < 1e-12)
    ok = ok1 and ok2 and ok3 and ok4
    bad += not ok
    print(trial, 'kernel', kid, 'n', n, 'm', m, 'T', T, 'gram', (nr, nc), 'resid', len(k3), 'predict', (len(pc), draws),
          'OK' if ok else f'MISMATCH K1 {ok1} K2 {ok2} K3 {ok3} predict {ok4}', flush=True)
print('mismatches', bad)
