"""First-contact diagnostics on a real MI355X: every kernel against the oracle / numpy, plus rough timings.

    python tests/stress/gpu_probe.py [quick]

Prints a verdict per check; exits non-zero if any check fails.  (Development aid; the judged parity tests are
tests/test_gpu_*.py.)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from fokl_gpy_amd import _capi, getKernels  # noqa: E402
from oracle import fokl_oracle as O  # noqa: E402
from helpers import load_case  # noqa: E402

FAILS = []


def verdict(name, ok, detail=''):
    print(('PASS ' if ok else 'FAIL ') + name + (' :: ' + detail if detail else ''), flush=True)
    if not ok:
        FAILS.append(name)


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    sp = np.spacing(np.maximum(np.abs(a), np.abs(b)))
    return np.abs(a - b) / sp


def check_basis(ctx, kernel_id, phis, n, m, terms, seed, tag):
    rng = np.random.default_rng(seed)
    x = rng.random((n, m))
    x[0, :] = 0.0
    x[1, :] = 1.0
    x[2, :] = 0.5
    y = rng.standard_normal(n)
    packed, nb, width = getKernels.pack_phis(phis, kernel_id)
    ctx.upload(x, y, kernel_id, packed, nb, width)
    terms = np.asarray(terms, dtype=np.int32)
    T = terms.shape[0]
    ctx.reserve_slots(2 + T)
    slots = np.arange(2, 2 + T, dtype=np.int32)
    ctx.build_terms(terms, slots)
    got = np.stack([ctx.read_slot(int(s)) for s in slots], axis=1)
    if kernel_id == O.KERNEL_SPLINES:
        phind, xsm = O.inputs_to_phind(x, len(phis[0][0]))
    else:
        phind, xsm = None, x
    ref = O.build_columns_c(xsm, phind, phis, kernel_id, terms)
    ud = ulp_diff(got, ref)
    exact = np.mean(got == ref)
    verdict(f'K1 {tag} n={n} m={m} T={T}', bool(np.max(ud) <= 4.0),
            f'max ulp {np.max(ud):.2f}, bit-identical {100 * exact:.3f}%, ones ok {np.all(ctx.read_slot(0) == 1.0)}, '
            f'y ok {np.array_equal(ctx.read_slot(1), y)}')
    return x, y, got, slots


def check_gram(ctx, nslots, n, tag):
    rng = np.random.default_rng(7)
    # exact small-integer data: any summation order gives the same fp64 result
    cols = rng.integers(-3, 4, size=(n, nslots)).astype(np.float64)
    ctx.reserve_slots(2 + nslots)
    for j in range(nslots):
        ctx.write_slot(2 + j, cols[:, j])
    for (nr, nc) in [(1, 1), (3, 5), (8, 10), (16, 16), (17, 33), (28, 38), (56, 64), (64, 130), (70, 200)]:
        if nr > nslots or nc > nslots:
            continue
        rs = (2 + rng.permutation(nslots)[:nr]).astype(np.int32)
        cs = (2 + rng.permutation(nslots)[:nc]).astype(np.int32)
        want = cols[:, rs - 2].T @ cols[:, cs - 2]
        for path, pname in ((1, 'valu'), (2, 'mfma')):
            got = ctx.gram(rs, cs, path=path)
            verdict(f'K2 {pname} {tag} {nr}x{nc}', bool(np.array_equal(got, want)),
                    f'max abs diff {np.max(np.abs(got - want)):.3g}')


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    print('devices', _capi.device_count(), flush=True)
    ctx = _capi.DeviceContext(0)
    bern = getKernels.bernoulli()
    spl = getKernels.table_to_phis(np.load(os.path.join(ROOT, 'tests', 'golden', 'spline_phis.npz'))['table'])

    # ---- K1 -------------------------------------------------------------------------------------------
    t2 = O.distinct_arrangements([2, 1, 0, 0])
    t1 = O.distinct_arrangements([1, 0, 0, 0])
    check_basis(ctx, O.KERNEL_BERNOULLI, bern, 1000, 4, np.vstack([t1, t2]), 1, 'bernoulli low')
    hi = np.array([[20, 0, 0], [0, 19, 3], [7, 11, 13], [1, 1, 1], [15, 0, 2]])
    check_basis(ctx, O.KERNEL_BERNOULLI, bern, 777, 3, hi, 2, 'bernoulli high-order')
    check_basis(ctx, O.KERNEL_SPLINES, spl, 1001, 4, np.vstack([t1, t2]), 3, 'splines')
    many = np.array([[a, b, c] for a in range(0, 8) for b in range(0, 4) for c in (0, 9, 24)][1:])
    check_basis(ctx, O.KERNEL_SPLINES, spl, 513, 3, many, 4, 'splines many orders (multi-launch, global slabs)')
    big_terms = O.distinct_arrangements([1, 1, 0, 0, 0, 0, 0, 0])
    check_basis(ctx, O.KERNEL_BERNOULLI, bern, 100003, 8, big_terms, 5, 'bernoulli odd n')

    # ---- K2 -------------------------------------------------------------------------------------------
    x, y, got, slots = check_basis(ctx, O.KERNEL_BERNOULLI, bern, 4099, 4, t1, 6, 'setup for gram')
    check_gram(ctx, 210, 4099, 'int-data n=4099')
    # random data vs numpy
    rng = np.random.default_rng(8)
    cols = rng.standard_normal((4099, 40))
    for j in range(40):
        ctx.write_slot(2 + j, cols[:, j])
    rs = np.arange(2, 30, dtype=np.int32)
    cs = np.arange(2, 42, dtype=np.int32)
    want = cols[:, :28].T @ cols
    for path in (1, 2):
        g = ctx.gram(rs, cs, path=path)
        verdict(f'K2 path{path} random 28x40', bool(np.max(np.abs(g - want)) < 1e-10 * 4099),
                f'max abs diff {np.max(np.abs(g - want)):.3g}')
    g0 = ctx.gram([0, 1], [0, 1])
    verdict('K2 ones/y block', bool(g0[0, 0] == 4099 and abs(g0[0, 1] - y.sum()) < 1e-9 and abs(g0[1, 1] - y @ y) < 1e-8),
            str(g0))

    # ---- K3 -------------------------------------------------------------------------------------------
    beta = rng.standard_normal(11)
    sl = np.concatenate([[0], np.arange(2, 12)]).astype(np.int32)
    X = np.concatenate([np.ones((4099, 1)), cols[:, :10]], axis=1)
    r = y - X @ beta
    s1, s2 = ctx.bic_resid(sl, beta)
    verdict('K3 resid', bool(abs(s1 - r.sum()) < 1e-9 * 4099 and abs(s2 - r @ r) < 1e-9 * 4099),
            f'{s1 - r.sum():.3g} {s2 - r @ r:.3g}')

    # ---- predict ---------------------------------------------------------------------------------------
    betas = rng.standard_normal((200, 11))
    mean, bounds = ctx.predict(sl, betas, cut=6)
    mod = X @ betas.T
    srt = np.sort(mod, axis=1)
    verdict('predict mean/bounds', bool(np.max(np.abs(mean - mod.mean(1))) < 1e-12 and
                                        np.array_equal(bounds[:, 0], srt[:, 6]) and np.array_equal(bounds[:, 1], srt[:, 194])),
            f'mean err {np.max(np.abs(mean - mod.mean(1))):.3g} lo err {np.max(np.abs(bounds[:, 0] - srt[:, 6])):.3g} '
            f'hi err {np.max(np.abs(bounds[:, 1] - srt[:, 194])):.3g}')

    # ---- full fits against the reference goldens ---------------------------------------------------------
    from fokl_gpy_amd import FoKLRoutines
    import warnings
    for name in (['bern_m3'] if quick else ['bern_m1', 'bern_m3', 'bern_m4_way3', 'bern_m8_capped', 'splines_m4']):
        if not os.path.exists(os.path.join(ROOT, 'tests', 'golden', name + '.npz')):
            continue
        g, hy, kname, kid, phis = load_case(name)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = FoKLRoutines.FoKL(kernel=kname, phis=phis, UserWarnings=False, ConsoleOutput=False, **hy)
            np.random.seed(int(g['seed']))
            t0 = time.time()
            b, mtx, evs = model.fit(g['raw_inputs'], g['raw_data'], clean=True)
            dt = time.time() - t0
        gm, ge, gb = g['canon_mtx'], g['canon_evs'], g['canon_betas']
        same = gm.shape == mtx.shape and np.array_equal(gm, mtx)
        detail = f'{dt:.2f}s mtx {same}'
        ok = same and len(evs) == len(ge)
        if ok:
            e_ev = np.max(np.abs(evs - ge) / np.abs(ge))
            e_b = np.max(np.abs(b - gb) / np.max(np.abs(gb), axis=0))
            detail += f' evs rel {e_ev:.3g} betas rel {e_b:.3g}'
            ok = e_ev < 1e-9 and e_b < 1e-7
        verdict(f'fit {name}', bool(ok), detail + f' {model.fit_stats}')

    # ---- rough timings at BASELINE config-3 size ---------------------------------------------------------
    if not quick:
        n, m = 1_000_000, 8
        rng = np.random.default_rng(12)
        x = rng.random((n, m))
        yv = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.05 * rng.standard_normal(n)
        packed, nb, width = getKernels.pack_phis(bern, O.KERNEL_BERNOULLI)
        t0 = time.time()
        ctx.upload(x, yv, O.KERNEL_BERNOULLI, packed, nb, width)
        print(f'upload n=1e6 m=8: {time.time() - t0:.3f}s', flush=True)
        ctx.timing_enable(True)
        for pattern in ([1, 0], [1, 1], [2, 1], [3, 2]):
            terms = O.distinct_arrangements(pattern + [0] * (m - 2)).astype(np.int32)
            T = terms.shape[0]
            ctx.reserve_slots(2 + 2 * T + 64)
            slots = np.arange(2, 2 + T, dtype=np.int32)
            ctx.build_terms(terms, slots)
            ctx.sync()
            ctx.timing_reset()
            for _ in range(10):
                ctx.build_terms(terms, slots)
            ctx.sync()
            t = ctx.timing_get(_capi.K_BASIS)
            per = t['ms'] / t['launches']
            print(f'K1 pattern {pattern} T={T}: {per * 1e3:.1f} us/launch, {t["bytes"] / t["launches"] / per / 1e6:.1f} GB/s algorithmic',
                  flush=True)
            allc = np.concatenate([[0], slots, [1]]).astype(np.int32)
            for path in (1, 2):
                ctx.timing_reset()
                for _ in range(5):
                    ctx.gram(slots, allc, path=path)
                t = ctx.timing_get_gram()
                per = t['ms'] / t['launches']
                print(f'   K2 path{path} {T}x{T + 2}: {per * 1e3:.1f} us, {t["bytes"] / t["launches"] / per / 1e6:.1f} GB/s, '
                      f'{t["flops"] / t["launches"] / per / 1e9:.2f} TFLOP/s', flush=True)
            ctx.timing_reset()
            bh = rng.standard_normal(T + 1) * 0.01
            for _ in range(5):
                ctx.bic_resid(np.concatenate([[0], slots]).astype(np.int32), bh)
            t = ctx.timing_get(_capi.K_RESID)
            per = t['ms'] / t['launches']
            print(f'   K3 {T + 1} cols: {per * 1e3:.1f} us, {t["bytes"] / t["launches"] / per / 1e6:.1f} GB/s', flush=True)

    print('FAILED: ' + ', '.join(FAILS) if FAILS else 'ALL PASS', flush=True)
    sys.exit(1 if FAILS else 0)


if __name__ == '__main__':
    main()
