"""Parity tests proper: the HIP path, called through the C ABI, against the oracle and the reference fixtures.

Tolerances:
  * K1 columns: splines bit-identical to the reference's scalar arithmetic; Bernoulli bit-identical wherever glibc's
    pow is correctly rounded (> 99.5 % of entries) and otherwise within 2^-50 * prod_k sum_j |c_j x^j| (one ulp
    of the largest monomial per factor) -- the kernel rounds x**j correctly, libm's pow is off by 1 ulp in ~0.08 %;
  * K2 / K3 / predict: exact on small-integer data (any summation order), <= 1e-12 relative on real data;
  * fits: selected interaction matrix exact, BIC trace <= 1e-9 relative, draws <= 1e-9 * max|column|.
"""
import os
import warnings

import numpy as np
import pytest

from helpers import GOLDEN, FIT_CASES, load_case
from fokl_gpy_amd import _capi, getKernels, FoKLRoutines, engine
from oracle import fokl_oracle as O

pytestmark = pytest.mark.gpu

BERN = getKernels.bernoulli()
SPL = getKernels.table_to_phis(np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table'])


def upload(ctx, x, y, kid):
    phis = SPL if kid == O.KERNEL_SPLINES else BERN
    packed, nb, width = getKernels.pack_phis(phis, kid)
    ctx.upload(x, y, kid, packed, nb, width)
    return phis


def build_and_read(ctx, terms):
    terms = np.asarray(terms, dtype=np.int32)
    T = terms.shape[0]
    ctx.reserve_slots(2 + T)
    slots = np.arange(2, 2 + T, dtype=np.int32)
    ctx.build_terms(terms, slots)
    return np.stack([ctx.read_slot(int(s)) for s in slots], axis=1)


def oracle_columns(x, kid, phis, terms):
    if kid == O.KERNEL_SPLINES:
        phind, xsm = O.inputs_to_phind(x, len(phis[0][0]))
    else:
        phind, xsm = None, x
    return O.build_columns_c(xsm, phind, phis, kid, np.asarray(terms, dtype=np.int32))


def bernoulli_bound(x, terms):
    """prod over the term's inputs of sum_j |c_j| |x|^j -- the magnitude one ulp of a monomial is measured against."""
    out = np.ones((x.shape[0], len(terms)))
    for j, term in enumerate(terms):
        for k, o in enumerate(term):
            if o:
                c = np.abs(np.asarray(BERN[o - 1]))
                out[:, j] *= sum(c[p] * np.abs(x[:, k]) ** p for p in range(len(c)))
    return out


# ---------------------------------------------------------------------------------------------------------
# K1 basis build
# ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 511, 512, 513, 4099])
def test_k1_ragged_row_counts(device_ctx, n):
    rng = np.random.default_rng(n)
    x = rng.random((n, 3))
    y = rng.standard_normal(n)
    for kid in (O.KERNEL_SPLINES, O.KERNEL_BERNOULLI):
        phis = upload(device_ctx, x, y, kid)
        terms = np.array([[1, 0, 0], [0, 2, 0], [1, 1, 0], [2, 0, 3], [1, 2, 3]])
        got = build_and_read(device_ctx, terms)
        want = oracle_columns(x, kid, phis, terms)
        assert got.shape == want.shape
        if kid == O.KERNEL_SPLINES:
            assert np.array_equal(got, want)
        else:
            assert np.all(np.abs(got - want) <= 2.0 ** -50 * bernoulli_bound(x, terms))
        assert np.all(device_ctx.read_slot(0) == 1.0) and np.array_equal(device_ctx.read_slot(1), y)


def test_k1_splines_bit_identical_including_knots_and_ends(device_ctx):
    rng = np.random.default_rng(1)
    knots = np.arange(0, 500) / 499.0
    col = np.concatenate([rng.random(3000), knots, [0.0, 1.0, 1e-300, 1 - 1e-16, 0.5, 1 / 499 + 1e-17, 2 / 499 - 1e-17]])
    x = np.stack([col, rng.permutation(col), col[::-1]], axis=1)
    phis = upload(device_ctx, x, np.zeros(x.shape[0]), O.KERNEL_SPLINES)
    terms = [[a, b, c] for a in (0, 1, 5, 24) for b in (0, 2, 9) for c in (0, 3)][1:]
    got = build_and_read(device_ctx, terms)           # > 4 distinct orders: exercises LDS-staged and global slabs
    assert np.array_equal(got, oracle_columns(x, O.KERNEL_SPLINES, phis, terms))


def test_k1_bernoulli_low_orders_almost_always_bit_identical(device_ctx):
    rng = np.random.default_rng(2)
    x = rng.random((20000, 8))
    x[:3] = [[0.0] * 8, [1.0] * 8, [0.5] * 8]
    upload(device_ctx, x, np.zeros(x.shape[0]), O.KERNEL_BERNOULLI)
    terms = np.vstack([engine.distinct_arrangements([1] + [0] * 7), engine.distinct_arrangements([2, 1] + [0] * 6),
                       engine.distinct_arrangements([3, 2] + [0] * 6)[:40]]).astype(np.int32)
    got = build_and_read(device_ctx, terms)
    want = oracle_columns(x, O.KERNEL_BERNOULLI, BERN, terms)
    assert np.mean(got == want) > 0.995
    assert np.all(np.abs(got - want) <= 2.0 ** -50 * bernoulli_bound(x, terms))


def test_k1_bernoulli_all_twenty_orders(device_ctx):
    rng = np.random.default_rng(3)
    x = rng.random((5000, 2))
    upload(device_ctx, x, np.zeros(5000), O.KERNEL_BERNOULLI)
    terms = np.array([[o, 0] for o in range(1, 21)] + [[20, 19], [7, 13], [1, 20]])
    got = build_and_read(device_ctx, terms)
    want = oracle_columns(x, O.KERNEL_BERNOULLI, BERN, terms)
    assert np.all(np.abs(got - want) <= 2.0 ** -49 * bernoulli_bound(x, terms))


def test_k1_many_terms_split_over_launches(device_ctx):
    """More distinct (input, order) factors than one LDS factor table holds -> several launches, same result."""
    rng = np.random.default_rng(4)
    x = rng.random((3001, 6))
    upload(device_ctx, x, np.zeros(3001), O.KERNEL_BERNOULLI)
    terms = np.array([[(i + k) % 9 if (i + k) % 3 else 0 for k in range(6)] for i in range(1, 140)])
    terms = terms[terms.sum(1) > 0]
    got = build_and_read(device_ctx, terms)
    want = oracle_columns(x, O.KERNEL_BERNOULLI, BERN, terms)
    assert np.all(np.abs(got - want) <= 2.0 ** -49 * bernoulli_bound(x, terms))


def test_k1_single_input(device_ctx):
    x = np.linspace(0, 1, 257)[:, None]
    upload(device_ctx, x, np.zeros(257), O.KERNEL_SPLINES)
    got = build_and_read(device_ctx, [[1], [2], [7]])
    assert np.array_equal(got, oracle_columns(x, O.KERNEL_SPLINES, SPL, [[1], [2], [7]]))


def test_argument_errors_are_reported(device_ctx):
    upload(device_ctx, np.random.default_rng(0).random((100, 2)), np.zeros(100), O.KERNEL_BERNOULLI)
    device_ctx.reserve_slots(8)
    with pytest.raises(_capi.FoklNativeError) as e:
        device_ctx.build_terms([[1, 0]], [1])                       # reserved slot
    assert e.value.code == -2
    with pytest.raises(_capi.FoklNativeError):
        device_ctx.build_terms([[21, 0]], [2])                      # order beyond the table
    with pytest.raises(_capi.FoklNativeError):
        device_ctx.build_terms([[0, 0]], [2])                       # empty term
    with pytest.raises(_capi.FoklNativeError):
        device_ctx.gram([0, 99999], [0])                            # slot out of range
    with pytest.raises(_capi.FoklNativeError):
        device_ctx.read_slot(2, 50, 100)                            # row range
    fresh = _capi.DeviceContext(device_ctx.device)
    with pytest.raises(_capi.FoklNativeError) as e:
        fresh.build_terms([[1, 0]], [2])                            # before upload
    assert e.value.code == -3
    # staged uploads (fit's normalisation on the device): nothing staged, or staged with another shape
    packed, nb, width = getKernels.pack_phis(getKernels.bernoulli(), O.KERNEL_BERNOULLI)
    with pytest.raises(_capi.FoklNativeError) as e:
        fresh._staged_shape = (100, 2)
        fresh.upload_staged(np.zeros(100), O.KERNEL_BERNOULLI, packed, nb, width, np.zeros(2), np.ones(2))
    assert e.value.code == -3
    fresh.stage_inputs(np.ascontiguousarray(np.random.default_rng(1).random((100, 2))))
    with pytest.raises(_capi.FoklNativeError):
        fresh._staged_shape = (50, 2)
        fresh.upload_staged(np.zeros(50), O.KERNEL_BERNOULLI, packed, nb, width, np.zeros(2), np.ones(2))
    with pytest.raises(ValueError):
        fresh.stage_inputs(np.zeros((10, 3), dtype=np.float32))
    with pytest.raises(_capi.FoklNativeError):
        fresh.download_inputs()                                      # no dataset on this context yet
    fresh.close()


# ---------------------------------------------------------------------------------------------------------
# K2 Gram, K3 residual, predict
# ---------------------------------------------------------------------------------------------------------

def load_columns(ctx, cols):
    n, k = cols.shape
    ctx.reserve_slots(2 + k)
    for j in range(k):
        ctx.write_slot(2 + j, cols[:, j])


def dev_kernels_built(ctx):
    """The retired Gram kernels (round-1 panels = path 3, 4x4x4 tile lists, third LDS-DMA buffer) are compiled only into
    development builds (make -C fokl_gpy_amd/csrc DEV=1): the product library answers path 3 with an argument error."""
    try:
        ctx.gram(np.array([0], dtype=np.int32), np.array([0, 1], dtype=np.int32), path=3)
        return True
    except _capi.FoklNativeError as exc:
        assert exc.code == -2 and 'development build' in str(exc)
        return False


@pytest.mark.parametrize('mfma4', ['default', '2', 'nodma'])          # default / the 4x4x4 form (dev builds) / no LDS-DMA kernel
@pytest.mark.parametrize('n', [1, 31, 32, 33, 1000, 4099, 120001])     # the last: workgroups loop over several chunks
def test_k2_exact_on_integer_data(device_ctx, n, mfma4, monkeypatch):
    rng = np.random.default_rng(n)
    upload(device_ctx, rng.random((n, 1)), rng.integers(-3, 4, n).astype(float), O.KERNEL_BERNOULLI)
    dev = dev_kernels_built(device_ctx)
    if mfma4 == 'nodma':
        if not dev:
            pytest.skip('A/B knobs of the Gram kernels are read by development builds only')
        monkeypatch.setenv('FOKL_GRAM_DMA', '0')
    elif mfma4 != 'default':
        if not dev:
            pytest.skip('the 4x4x4 tile-list kernel is compiled into development builds only')
        monkeypatch.setenv('FOKL_GRAM_MFMA4', mfma4)
    paths = (0, 1, 2, 3) if dev else (0, 1, 2)
    cols = rng.integers(-3, 4, size=(n, 150)).astype(np.float64)
    load_columns(device_ctx, cols)
    for nr, nc in [(1, 1), (2, 3), (8, 10), (16, 16), (17, 33), (28, 38), (56, 66), (65, 131), (70, 150)]:
        rs = (2 + rng.permutation(150)[:nr]).astype(np.int32)
        cs = (2 + rng.permutation(150)[:nc]).astype(np.int32)
        want = cols[:, rs - 2].T @ cols[:, cs - 2]
        for path in paths:
            assert np.array_equal(device_ctx.gram(rs, cs, path=path), want), (nr, nc, path)
        # the search's own pattern: the row-side columns reappear in the middle of the column list, so the tile-list
        # kernel computes their square once (tiles on or above the diagonal) and mirrors the rest
        if nc > nr:
            rest = np.setdiff1d(np.arange(2, 152), rs)
            cs2 = np.concatenate([[0], rest[:nc - nr - 2], rs, [1]])[:nc].astype(np.int32) if nc - nr >= 2 else \
                np.concatenate([rs, rest[:nc - nr]]).astype(np.int32)
            ref = np.concatenate([np.ones((n, 1)), device_ctx.read_slot(1)[:, None], cols], axis=1)
            want2 = ref[:, rs].T @ ref[:, cs2]
            for path in paths[2:]:
                assert np.array_equal(device_ctx.gram(rs, cs2, path=path), want2), (nr, nc, path, 'symmetric')
    g = device_ctx.gram([0, 1, 2], [0, 1, 2])
    y = device_ctx.read_slot(1)
    assert g[0, 0] == n and g[0, 1] == y.sum() and g[1, 1] == y @ y and g[0, 2] == cols[:, 0].sum()


def test_k2_paths_agree_and_are_reproducible(device_ctx, monkeypatch):
    rng = np.random.default_rng(9)
    n = 50000
    upload(device_ctx, rng.random((n, 1)), rng.standard_normal(n), O.KERNEL_BERNOULLI)
    cols = rng.standard_normal((n, 70)) * np.exp(rng.standard_normal(70))
    load_columns(device_ctx, cols)
    rs = np.arange(2, 58, dtype=np.int32)
    cs = np.concatenate([[0], np.arange(2, 72), [1]]).astype(np.int32)
    full = np.concatenate([np.ones((n, 1)), cols, device_ctx.read_slot(1)[:, None]], axis=1)
    want = cols[:, :56].T @ full
    scale = np.sqrt(np.outer(np.sum(cols[:, :56] ** 2, 0), np.sum(full ** 2, 0)))
    g1 = device_ctx.gram(rs, cs, path=1)
    g2 = device_ctx.gram(rs, cs, path=2)
    assert np.max(np.abs(g1 - want) / scale) < 1e-13 and np.max(np.abs(g2 - want) / scale) < 1e-13
    assert np.array_equal(g2, device_ctx.gram(rs, cs, path=2))          # fixed-order reduction: bitwise repeatable
    assert np.array_equal(g1, device_ctx.gram(rs, cs, path=1))
    # the 56 row-side columns leave a ragged last row tile: its tiles are formed as 8 x 16 half tiles on the 4x4x4 MFMA
    # (half-tile slots) -- same products, same order of summation per element, the same bits as whole tiles
    assert _capi.gram_plan(rs, cs)['half'].sum() > 0
    if dev_kernels_built(device_ctx):                                   # (the knob exists in development builds only)
        monkeypatch.setenv('FOKL_GRAM_HALF', '0')
        assert _capi.gram_plan(rs, cs)['half'].sum() == 0
        assert np.array_equal(g2, device_ctx.gram(rs, cs, path=2))
        monkeypatch.delenv('FOKL_GRAM_HALF')
    # the launch / fetch pair returns the blocking call's block whatever is launched in between (other Gram blocks and
    # residual passes use other result buffers); a launch drops a block that was never fetched
    auto = device_ctx.gram(rs, cs)
    small = device_ctx.gram(rs[:3], cs[:9])
    device_ctx.gram_launch(rs[:5], cs[:7])
    shape = device_ctx.gram_launch(rs, cs)
    beta = rng.standard_normal(cs.shape[0] - 1)
    moments = device_ctx.bic_resid(cs[:-1], beta)
    device_ctx.bic_resid_launch(cs[:-1], beta)
    assert np.array_equal(device_ctx.gram(rs[:3], cs[:9]), small)
    assert np.array_equal(device_ctx.gram_fetch(shape), auto)
    assert device_ctx.bic_resid_fetch() == moments
    with pytest.raises(_capi.FoklNativeError):
        device_ctx.gram_fetch(shape)                                     # nothing on its way any more
    # round 6: the native sub-stage loop queues the coming sub-stage's Gram launch BEHIND the model's residual pass; the
    # pass's fetch waits for an event of its own (not for the stream) and returns the pass's moments, the block follows
    device_ctx.bic_resid_launch(cs[:-1], beta)
    shape = device_ctx.gram_launch(rs, cs)
    assert device_ctx.bic_resid_fetch() == moments
    assert np.array_equal(device_ctx.gram_fetch(shape), auto)
    device_ctx.gram_launch(rs[:2], cs[:4])
    with pytest.raises(_capi.FoklNativeError):
        device_ctx.gram_fetch((3, 4))                                    # not the block that was launched
    np.testing.assert_allclose(device_ctx.gram_fetch((2, 4)), auto[:2, :4], rtol=1e-12, atol=1e-9)


def test_k3_residual_moments(device_ctx):
    rng = np.random.default_rng(10)
    for n in (1, 513, 20001):
        y = rng.standard_normal(n)
        upload(device_ctx, rng.random((n, 1)), y, O.KERNEL_BERNOULLI)
        cols = rng.standard_normal((n, 40))
        load_columns(device_ctx, cols)
        beta = rng.standard_normal(41)
        sl = np.concatenate([[0], np.arange(2, 42)]).astype(np.int32)
        r = y - (beta[0] + cols @ beta[1:])
        s1, s2 = device_ctx.bic_resid(sl, beta)
        assert abs(s1 - r.sum()) <= 1e-12 * np.abs(r).sum() + 1e-300
        assert abs(s2 - r @ r) <= 1e-12 * (r @ r)
        device_ctx.bic_resid_launch(sl, beta)                           # split form gives the same numbers
        assert device_ctx.bic_resid_fetch() == (s1, s2)
        s1z, s2z = device_ctx.bic_resid([0], [0.0])                     # linearity anchor: beta = 0 -> moments of y
        assert abs(s1z - y.sum()) <= 1e-12 * np.abs(y).sum() and abs(s2z - y @ y) <= 1e-12 * (y @ y)


@pytest.mark.parametrize('kid', [O.KERNEL_BERNOULLI, O.KERNEL_SPLINES])
def test_k3_matrix_free_agrees_with_the_stored_column_pass(device_ctx, kid):
    """fokl_bic_resid_terms_launch re-forms the model's distinct factors from the inputs with the operations of
    fokl_build_terms and evaluates the fit as their quadratic form: the same moments as the stored-column pass up to
    the association of the sum (1e-13 of the moments' scale; the oracle's columns to 1e-11) -- for every slot layout
    (inputs x orders per input: 8 x 1, 16 x 1, 8 x 2, 4 x 4, 2 x 8, 8 x 4, 16 x 2, 4 x 8), ragged row counts, one- and two-way terms,
    duplicated terms, subsets of a sub-stage's terms; models outside the layouts are refused, not overrun."""
    rng = np.random.default_rng(14)
    # (rows, inputs in the dataset, inputs used, orders per input used, highest order, terms)
    for n, m, used, per_input, top, n_terms in ((1, 8, 3, 1, 2, 5), (777, 8, 8, 1, 1, 36), (20001, 8, 8, 2, 4, 60),
                                                (65536 + 3, 16, 16, 1, 3, 120), (5000, 8, 4, 4, 6, 80),
                                                (3001, 8, 2, 8, 8, 25), (4099, 8, 8, 2, 2, 128), (9001, 8, 8, 4, 4, 200),
                                                (2500, 16, 16, 2, 3, 150), (1300, 8, 4, 8, 8, 90), (7000, 8, 8, 3, 5, 110)):
        x = rng.random((n, m))
        y = rng.standard_normal(n)
        phis = upload(device_ctx, x, y, kid)
        inputs = rng.choice(m, size=used, replace=False)
        orders = {int(k): rng.choice(np.arange(1, top + 1), size=min(per_input, top), replace=False) for k in inputs}
        terms = np.zeros((n_terms, m), dtype=np.int32)
        for j in range(n_terms):
            ways = int(rng.integers(1, 3)) if used > 1 else 1
            for k in rng.choice(inputs, size=ways, replace=False):
                terms[j, k] = int(rng.choice(orders[int(k)]))
        device_ctx.reserve_slots(2 + n_terms)
        slots = np.arange(2, 2 + n_terms, dtype=np.int32)
        device_ctx.build_terms(terms, slots)
        beta = rng.standard_normal(n_terms + 1)
        cols = oracle_columns(x, kid, phis, terms)
        for keep in (np.arange(n_terms), np.sort(rng.choice(n_terms, size=max(1, n_terms // 3), replace=False))):
            b = np.concatenate([beta[:1], beta[1 + keep]])
            want = device_ctx.bic_resid(np.concatenate([[0], slots[keep]]).astype(np.int32), b)
            device_ctx.bic_resid_terms_launch(terms[keep], b)
            got = device_ctx.bic_resid_fetch()
            r = y - (b[0] + cols[:, keep] @ b[1:])
            scale1, scale2 = np.abs(r).sum() + 1e-300, r @ r
            assert abs(got[0] - want[0]) <= 1e-13 * scale1 and abs(got[1] - want[1]) <= 1e-13 * scale2, (n, got, want)
            assert abs(got[0] - r.sum()) <= 1e-11 * scale1 and abs(got[1] - r @ r) <= 1e-11 * scale2
            device_ctx.bic_resid_terms_launch(terms[keep], b)            # the same launch again: the same bits
            assert device_ctx.bic_resid_fetch() == got
    # intercept-only model, and the limits are reported, not overrun
    device_ctx.bic_resid_terms_launch(np.zeros((0, m), dtype=np.int32), [0.25])
    s1, s2 = device_ctx.bic_resid_fetch()
    assert abs(s1 - (y - 0.25).sum()) <= 1e-12 * np.abs(y - 0.25).sum()
    three_way = np.zeros((1, m), dtype=np.int32)
    three_way[0, :3] = 1
    too_deep = np.zeros((40, m), dtype=np.int32)                          # 8 inputs x 5 orders: no layout
    for j in range(40):
        too_deep[j, j % 8] = 1 + j // 8
    for bad in (three_way, too_deep):
        with pytest.raises(_capi.FoklNativeError):
            device_ctx.bic_resid_terms_launch(bad, np.zeros(bad.shape[0] + 1))


def test_device_dgemm_is_the_host_dgemm_to_rounding(device_ctx):
    """fokl_device_dgemm (the eigen-update's product on the matrix cores, BLAS dgemm's signature): C = A B for column-major
    operands with leading dimensions larger than the matrices, ragged sizes on either side of the 64 x 64 x 16 tiling, within
    a few ulp of |A| |B| of numpy's product; calls it does not take (a transpose, beta != 0, small sizes) reach the host
    dgemm it wraps and return ITS bits; the counters say which way each call went."""
    import ctypes
    lib = _capi.load()
    host = _capi._scipy_dgemm_address()
    assert host
    entry = ctypes.c_void_p(0)
    assert lib.fokl_device_dgemm_configure(0, ctypes.c_void_p(host), 48, ctypes.byref(entry)) == 0 and entry.value
    proto = ctypes.CFUNCTYPE(None, *([ctypes.c_void_p] * 13))
    device_dgemm, host_dgemm = proto(entry.value), proto(host)

    def call(fn, ta, tb, a, b, c, m, n, k, alpha=1.0, beta=0.0):
        args = [ctypes.c_char(ta), ctypes.c_char(tb), ctypes.c_int(m), ctypes.c_int(n), ctypes.c_int(k), ctypes.c_double(alpha),
                None, ctypes.c_int(a.shape[1]), None, ctypes.c_int(b.shape[1]), ctypes.c_double(beta), None, ctypes.c_int(c.shape[1])]
        ptr = lambda v: ctypes.cast(ctypes.byref(v), ctypes.c_void_p)
        fn(ptr(args[0]), ptr(args[1]), ptr(args[2]), ptr(args[3]), ptr(args[4]), ptr(args[5]), a.ctypes.data, ptr(args[7]),
           b.ctypes.data, ptr(args[9]), ptr(args[10]), c.ctypes.data, ptr(args[12]))

    def counters():
        v = [ctypes.c_int64(0) for _ in range(3)]
        assert lib.fokl_device_dgemm_stats(*[ctypes.byref(x) for x in v]) == 0
        return [x.value for x in v]

    rng = np.random.default_rng(21)
    for m, n, k in ((16, 48, 48), (64, 64, 64), (65, 129, 67), (292, 585, 586), (293, 585, 586), (100, 200, 50), (17, 49, 1000)):
        # column-major with padding: array[col, row], leading dimension = shape[1]
        a = rng.standard_normal((k, m + 3))
        b = rng.standard_normal((n, k + 5))
        c = np.full((n, m + 2), np.nan)
        before = counters()
        call(device_dgemm, b'N', b'N', a, b, c, m, n, k)
        after = counters()
        assert after[0] == before[0] + 1 and after[1] == before[1] + 1 and after[2] == before[2]
        want = b[:, :k] @ a[:, :m]                                        # (C' = B' A' in this layout)
        bound = np.abs(b[:, :k]) @ np.abs(a[:, :m])
        assert np.all(np.abs(c[:, :m] - want) <= 8 * np.finfo(float).eps * bound)
        assert np.all(np.isnan(c[:, m:]))                                 # nothing written beyond the m rows
    # what it does not take goes to the host routine: the same bits as calling that directly
    a, b = rng.standard_normal((40, 40)), rng.standard_normal((40, 40))
    for ta, tb, m, n, k, beta in ((b'T', b'N', 40, 40, 40, 0.0), (b'N', b'N', 40, 40, 40, 0.5), (b'N', b'N', 24, 24, 24, 0.0),
                                  (b'N', b'N', 8, 40, 40, 0.0)):
        c1, c2 = np.ones((40, 40)), np.ones((40, 40))
        before = counters()
        call(device_dgemm, ta, tb, a, b, c1, m, n, k, beta=beta)
        call(host_dgemm, ta, tb, a, b, c2, m, n, k, beta=beta)
        assert counters()[1] == before[1] and np.array_equal(c1, c2)


def test_predict_mean_and_order_statistics(device_ctx):
    rng = np.random.default_rng(11)
    n = 3000
    upload(device_ctx, rng.random((n, 1)), np.zeros(n), O.KERNEL_BERNOULLI)
    cols = rng.standard_normal((n, 12))
    load_columns(device_ctx, cols)
    sl = np.concatenate([[0], np.arange(2, 14)]).astype(np.int32)
    X = np.concatenate([np.ones((n, 1)), cols], axis=1)
    for draws in (40, 1000):
        betas = rng.standard_normal((draws, 13))
        cut = int(np.floor(draws * 0.025) + 1)
        mean, bounds = device_ctx.predict(sl, betas, cut)
        mod = X @ betas.T
        srt = np.sort(mod, axis=1)
        assert np.max(np.abs(mean - mod.mean(1))) < 1e-12
        assert np.max(np.abs(bounds[:, 0] - srt[:, cut])) < 1e-12 and np.max(np.abs(bounds[:, 1] - srt[:, draws - cut])) < 1e-12
        assert np.max(np.abs(device_ctx.predict(sl, betas) - mean)) < 1e-13        # two kernels, same numbers


def test_predict_bounds_over_thousands_of_draws(device_ctx):
    """evaluate(ReturnBounds=True) / coverage3 over >= 5 080 draws need the 128-th and later order
    statistics per row, more than the on-chip lists hold; the reference sorts any number of draws (FR:971-977).  The
    lists then live in device memory: same result as numpy's sort, here with 6 200 and 12 000 draws."""
    rng = np.random.default_rng(23)
    n, nc = 700, 9
    upload(device_ctx, rng.random((n, 1)), np.zeros(n), O.KERNEL_BERNOULLI)
    cols = rng.standard_normal((n, nc - 1))
    load_columns(device_ctx, cols)
    X = np.concatenate([np.ones((n, 1)), cols], axis=1)
    sl = np.concatenate([[0], np.arange(2, 2 + nc - 1)]).astype(np.int32)
    for draws in (6200, 12000):
        betas = rng.standard_normal((draws, nc)) * np.linspace(1.0, 0.1, nc)
        cut = int(np.floor(draws * 0.025) + 1)
        assert cut + 1 > 128
        mean, bounds = device_ctx.predict(sl, betas, cut)
        mod = X @ betas.T
        srt = np.sort(mod, axis=1)
        scale = np.max(np.abs(mod))
        assert np.max(np.abs(mean - mod.mean(axis=1))) < 1e-12 * scale
        assert np.max(np.abs(bounds[:, 0] - srt[:, cut])) < 1e-12 * scale
        assert np.max(np.abs(bounds[:, 1] - srt[:, draws - cut])) < 1e-12 * scale


def test_predict_bounds_when_the_draws_are_far_from_gaussian(device_ctx):
    """The matrix-pipe predict kernel looks for the bounds among the predictions beyond mean -/+ z sigma of the row;
    draws with a few wild outliers inflate sigma until hardly anything passes, equal draws leave sigma = 0: both must
    end in the exact fallback and give numpy's order statistics."""
    rng = np.random.default_rng(17)
    n = 1237
    upload(device_ctx, rng.random((n, 1)), np.zeros(n), O.KERNEL_BERNOULLI)
    cols = rng.standard_normal((n, 6))
    load_columns(device_ctx, cols)
    sl = np.concatenate([[0], np.arange(2, 8)]).astype(np.int32)
    X = np.concatenate([np.ones((n, 1)), cols], axis=1)
    draws = 333
    cut = int(np.floor(draws * 0.025) + 1)
    wild = 0.01 * rng.standard_normal((draws, 7))
    wild[::41] *= 1e4
    same = np.tile(rng.standard_normal((1, 7)), (draws, 1))
    skew = np.exp(2.0 * rng.standard_normal((draws, 7)))
    for betas in (wild, same, skew):
        mean, bounds = device_ctx.predict(sl, betas, cut)
        mod = X @ betas.T
        srt = np.sort(mod, axis=1)
        scale = np.abs(mod).max(axis=1) + 1e-300
        assert np.max(np.abs(mean - mod.mean(1)) / scale) < 1e-12
        assert np.max(np.abs(bounds[:, 0] - srt[:, cut]) / scale) < 1e-12
        assert np.max(np.abs(bounds[:, 1] - srt[:, draws - cut]) / scale) < 1e-12


# ---------------------------------------------------------------------------------------------------------
# whole fits through the HIP backend vs the reference fixtures
# ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('name', [c for c in FIT_CASES if not c.startswith('testdata10')])
def test_fit_matches_reference_on_gpu(name):
    if not os.path.exists(os.path.join(GOLDEN, name + '.npz')):
        pytest.skip('fixture not generated')
    g, hy, kname, kid, phis = load_case(name)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=kname, phis=phis, UserWarnings=False, ConsoleOutput=False, **hy)
        np.random.seed(int(g['seed']))
        betas, mtx, evs = model.fit(g['raw_inputs'], g['raw_data'], clean=True)
        assert mtx.shape == g['canon_mtx'].shape and np.array_equal(mtx, g['canon_mtx'])
        assert len(evs) == len(g['canon_evs'])
        assert np.max(np.abs(evs - g['canon_evs']) / np.abs(g['canon_evs'])) < 1e-9
        gb = g['canon_betas']
        tol = 1e-6 if name == 'sigmoid_splines' else 1e-9
        assert np.max(np.abs(betas - gb) / np.max(np.abs(gb), axis=0)) < tol
        assert [t['cols'] for t in model.fit_trace] == g['canon_gibbs_sizes'].tolist()
        if 'canon_cov_mean' in g.files:
            mean, bounds, rmse = model.coverage3()
            scale = np.max(np.abs(g['canon_cov_mean']))
            ctol = 1e-6 if name == 'sigmoid_splines' else 1e-10
            assert np.array_equal(model.setnos, g['canon_setnos'])
            assert np.max(np.abs(mean - g['canon_cov_mean'])) < ctol * scale
            assert np.max(np.abs(bounds - g['canon_cov_bounds'])) < ctol * scale
            assert abs(rmse - float(g['canon_cov_rmse'])) < 1e-9


@pytest.mark.parametrize('name', ['testdata10_default', 'testdata10_changed'])
def test_reference_test_dataset_on_gpu(name):
    """The reference's own 10-row test set with both of its hyper-parameter sets (test/testdatatest.csv, seeds of
    test/makingdata.py).  With 10 rows the model saturates (P + 1 >= N) after 7 sub-stages and XtX becomes numerically
    singular; from there the reference's numbers are rounding noise of its own BLAS.  Pinned on
    everything before that point: BIC trace and the sequence of model evaluations."""
    g, hy, kname, kid, phis = load_case(name)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=kname, phis=phis, UserWarnings=False, ConsoleOutput=False, **hy)
        np.random.seed(int(g['seed']))
        betas, mtx, evs = model.fit(g['raw_inputs'], g['raw_data'], clean=True)
    assert np.max(np.abs(evs[:7] - g['canon_evs'][:7]) / np.abs(g['canon_evs'][:7])) < 1e-9
    sizes = g['canon_gibbs_sizes'].tolist()
    upto = sizes.index(10) if 10 in sizes else len(sizes)
    assert [t['cols'] for t in model.fit_trace][:upto] == sizes[:upto]
    assert betas.shape[0] == 1000 and mtx.shape[1] == 2


# ---------------------------------------------------------------------------------------------------------
# BASELINE sizes: size-independent properties at N = 1e6, M = 8
# ---------------------------------------------------------------------------------------------------------

def test_full_size_properties(device_ctx):
    rng = np.random.default_rng(12)
    n, m = 1_000_000, 8
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.05 * rng.standard_normal(n)
    upload(device_ctx, x, y, O.KERNEL_BERNOULLI)
    terms = np.vstack([engine.distinct_arrangements([1] + [0] * 7),
                       engine.distinct_arrangements([2, 1] + [0] * 6)]).astype(np.int32)
    T = terms.shape[0]
    device_ctx.reserve_slots(2 + T)
    slots = np.arange(2, 2 + T, dtype=np.int32)
    device_ctx.build_terms(terms, slots)
    # (1) random row sample of every column against the oracle
    pick = np.sort(rng.choice(n, 4000, replace=False))
    want = O.build_columns_c(np.ascontiguousarray(x[pick]), None, BERN, O.KERNEL_BERNOULLI, terms)
    got = np.stack([device_ctx.read_slot(int(s))[pick] for s in slots[:8]], axis=1)
    assert np.all(np.abs(got - want[:, :8]) <= 2.0 ** -50 * bernoulli_bound(x[pick], terms[:8]))
    blockrows = np.stack([device_ctx.read_slot(int(s), 123456, 2048) for s in slots], axis=1)
    want_b = O.build_columns_c(np.ascontiguousarray(x[123456:123456 + 2048]), None, BERN, O.KERNEL_BERNOULLI, terms)
    assert np.all(np.abs(blockrows - want_b) <= 2.0 ** -50 * bernoulli_bound(x[123456:123456 + 2048], terms))
    # (2) checksum of checksums: column sums via the Gram kernels == sums of the downloaded columns
    allc = np.concatenate([[0], slots, [1]]).astype(np.int32)
    g_mfma = device_ctx.gram(slots, allc, path=2)
    g_valu = device_ctx.gram(slots[:8], allc, path=1)
    col0 = device_ctx.read_slot(int(slots[0]))
    col5 = device_ctx.read_slot(int(slots[5]))
    assert abs(g_mfma[0, 0] - col0.sum()) <= 1e-12 * np.abs(col0).sum()
    assert abs(g_mfma[5, 1 + 5] - col5 @ col5) <= 1e-12 * (col5 @ col5)
    assert abs(g_mfma[0, -1] - col0 @ y) <= 1e-12 * np.abs(col0 * y).sum()
    # (3) the two Gram paths agree, and the new-vs-new block is symmetric
    scale = np.sqrt(np.outer(np.diag(g_mfma[:, 1:1 + T]), np.concatenate([[n], np.diag(g_mfma[:, 1:1 + T]), [y @ y]])))
    assert np.max(np.abs(g_mfma[:8] - g_valu) / scale[:8]) < 1e-13
    sq = g_mfma[:, 1:1 + T]
    assert np.max(np.abs(sq - sq.T) / scale[:, 1:1 + T]) < 1e-13
    # (4) residual linearity: r(beta) moments from K3 == those implied by the Gram blocks
    beta = np.zeros(T + 1)
    beta[0] = y.mean()
    beta[1:9] = 0.01 * rng.standard_normal(8)
    s1, s2 = device_ctx.bic_resid(np.concatenate([[0], slots]).astype(np.int32), beta)
    gfull = device_ctx.gram(np.concatenate([[0], slots[:8]]).astype(np.int32),
                            np.concatenate([[0], slots[:8], [1]]).astype(np.int32))
    b9 = beta[:9]
    s1_gram = y.sum() - gfull[0, :9] @ b9
    s2_gram = y @ y - 2 * b9 @ gfull[:, 9] + b9 @ gfull[:, :9] @ b9
    assert abs(s1 - s1_gram) <= 1e-9 * n and abs(s2 - s2_gram) <= 1e-9 * s2


# ---------------------------------------------------------------------------------------------------------
# RCCL plumbing (world of one on the single-GPU box; the N > 1 logic is covered by tests/test_dist_gloo.py)
# ---------------------------------------------------------------------------------------------------------

def test_rccl_world_of_one():
    from fokl_gpy_amd import dist
    ctx = _capi.DeviceContext(0)
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    comm = dist.RcclComm(ctx, 0, 1, unique_id=uid)
    v = np.array([1.5, -2.0, 3.25])
    assert np.array_equal(comm.allgather(v), v[None, :])
    assert np.array_equal(comm.allreduce_sum(v), v)
    comm.barrier()
    upload(ctx, np.random.default_rng(0).random((1000, 1)), np.ones(1000), O.KERNEL_BERNOULLI)
    assert ctx.gram([0], [0, 1], allreduce=True).tolist() == [[1000.0, 1000.0]]
    comm.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------
# whole fits at BASELINE sizes against the oracle (same seed; the reference itself needs hours at these N)
# ---------------------------------------------------------------------------------------------------------

def _fit_both(x, y, kname, kid, phis, seed, **hy):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel=kname, phis=phis, UserWarnings=False, ConsoleOutput=False, **hy)
        np.random.seed(seed)
        betas, mtx, evs = model.fit(x, y, clean=True)
        end_gpu = np.random.get_state()
        np.random.seed(seed)
        ob, om, oe = O.fit(model.inputs, model.data, phis, kid, eigh=O.eigh_canonical, **hy)
        end_cpu = np.random.get_state()
    assert om.shape == mtx.shape and np.array_equal(om, mtx)
    assert len(oe) == len(evs) and np.max(np.abs(oe - evs) / np.abs(oe)) < 1e-9
    assert np.max(np.abs(ob - betas) / np.max(np.abs(ob), axis=0)) < 1e-9
    assert np.array_equal(end_gpu[1], end_cpu[1]) and end_gpu[2:] == end_cpu[2:]
    return model


def test_config1_size_splines_fit_against_oracle():
    """BASELINE configs[1]: N = 1e5, M = 4, Cubic Splines (draws shortened so that the oracle's Python chain stays short)."""
    rng = np.random.default_rng(11)
    n, m = 100_000, 4
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.3 * x[:, 3] ** 2 + 0.05 * rng.standard_normal(n)
    model = _fit_both(x, y, 'Cubic Splines', O.KERNEL_SPLINES, SPL, 41, burnin=150, draws=150)
    assert model.mtx.shape[0] >= 4


def test_config2_size_bernoulli_capped_fit_against_oracle():
    """BASELINE configs[2] at full N = 1e6, M = 8 with the phis[:2] cap of SURVEY 8(d) (sub-stages of 8, 28, 8 terms)."""
    rng = np.random.default_rng(12)
    n, m = 1_000_000, 8
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.3 * x[:, 3] ** 2 + 0.5 * x[:, 4] * x[:, 5] + 0.05 * rng.standard_normal(n)
    model = _fit_both(x, y, 'Bernoulli Polynomials', O.KERNEL_BERNOULLI, BERN[:2], 1000, burnin=100, draws=100)
    assert model.fit_stats['terms_physical'] == 44


def test_many_inputs_three_way_fit_against_oracle():
    """configs[3] family: 3-way interactions with M = 12 inputs -- beyond what the reference's M! enumeration can run
    (SURVEY 8(a) F4), so the oracle (direct enumerator) is the only checker."""
    rng = np.random.default_rng(13)
    n, m = 4000, 12
    x = rng.random((n, m))
    y = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2] * x[:, 3] + 0.05 * rng.standard_normal(n)
    model = _fit_both(x, y, 'Bernoulli Polynomials', O.KERNEL_BERNOULLI, BERN[:3], 77, burnin=60, draws=60, way3=True,
                      tolerance=1)
    assert model.mtx.shape[1] == 12


def test_kill_test_bic_from_gram_agrees_with_the_device_pass(monkeypatch):
    """Kill-test candidates take their residual moments from the sub-stage's Gram instead of a K3 pass.
    FOKL_KILL_BIC=check runs both on every candidate of a full-size fit (N = 1e6: the cancellation in
    y'y - 2 b'Xty + b'XtX b is at its worst there) and records the largest relative disagreement of the BIC."""
    monkeypatch.setenv('FOKL_NOISE_PIPELINE', '1')                 # these are features of the threaded search
    rng = np.random.default_rng(12)
    n, m = 1_000_000, 8
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.3 * x[:, 3] ** 2 + 0.5 * x[:, 4] * x[:, 5] + 0.05 * rng.standard_normal(n)
    runs = {}
    for mode in ('check', 'gram'):
        monkeypatch.setenv('FOKL_KILL_BIC', mode)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = FoKLRoutines.FoKL(kernel='Bernoulli Polynomials', phis=BERN[:3], burnin=100, draws=100,
                                      UserWarnings=False, ConsoleOutput=False)
            np.random.seed(5)
            betas, mtx, evs = model.fit(x, y, clean=True)
        runs[mode] = (betas, mtx, evs, dict(model.fit_stats), np.random.get_state())
    st = runs['check'][3]
    assert st['kill_tests'] > 50 and st['bic_from_gram'] == 0
    assert 0 < st['bic_gram_max_rel'] < 1e-10, st['bic_gram_max_rel']
    assert runs['gram'][3]['bic_from_gram'] == runs['gram'][3]['kill_tests'] > 50
    assert np.array_equal(runs['gram'][1], runs['check'][1])
    np.testing.assert_allclose(runs['gram'][2], runs['check'][2], rtol=1e-10)
    np.testing.assert_allclose(runs['gram'][0], runs['check'][0], rtol=1e-7, atol=1e-9)
    assert np.array_equal(runs['gram'][4][1], runs['check'][4][1]) and runs['gram'][4][2:] == runs['check'][4][2:]


def test_model_saved_by_the_reference_evaluates_on_the_device():
    """a .fokl file written by the reference's own ``save`` loads as this package's class and its
    ``evaluate`` (K1 + predict kernel) reproduces the reference's numbers (fixture: make_golden.py saved_model)."""
    want = np.load(os.path.join(GOLDEN, 'ref_saved_model_expected.npz'))
    model = FoKLRoutines.load(os.path.join(GOLDEN, 'ref_saved_model.fokl'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        np.random.seed(6)
        mean, bounds = model.evaluate(want['inputs'], clean=True, ReturnBounds=True)
    np.testing.assert_allclose(mean, want['mean'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(bounds, want['bounds'], rtol=1e-11, atol=1e-11)


def test_device_probes_report_plausible_rates(device_ctx):
    """fokl_probe: what the device sustains for trivial streaming kernels and register-only fp64 MFMA (bench.py reports
    them beside the roofline fractions).  Loose physical bounds only."""
    read, write, mix, mfma = (device_ctx.probe(k) for k in range(4))
    # (the write probe streams in K1's shape with non-temporal stores: it can beat the read + write mix)
    assert 2e12 < write < read < 8.5e12 and 2e12 < mix < read
    assert 2e13 < mfma < 8e13
    with pytest.raises(_capi.FoklNativeError):
        device_ctx.probe(9)


def test_column_bounds_on_the_device_are_numpys(device_ctx):
    """fokl_stage_inputs: np.min / np.max of every column, exact and NaN-propagating, for ragged sizes."""
    rng = np.random.default_rng(5)
    for n, m in ((1, 1), (7, 3), (300_001, 5), (70_000, 33), (4096, 64)):
        x = rng.standard_normal((n, m)) * 10.0 ** rng.uniform(-3, 3, m)
        if n > 6:
            x[3, 0] = np.inf
            x[5, m - 1] = -np.inf
            if m > 2:
                x[n // 2, 1] = np.nan
        lows, highs = device_ctx.stage_inputs(np.ascontiguousarray(x))
        with np.errstate(invalid='ignore'):
            assert np.array_equal(lows, np.min(x, axis=0), equal_nan=True), (n, m)
            assert np.array_equal(highs, np.max(x, axis=0), equal_nan=True), (n, m)


def test_fit_normalises_raw_inputs_on_the_device_bit_for_bit(monkeypatch, tmp_path):
    """fit(clean=True) of a large float64 dataset: minima / maxima and (x - min) / (max - min) on the device
    (fokl_stage_inputs, fokl_upload_staged) instead of two host passes -- the same minmax, the same normalised numbers bit
    for bit (FR:395, 436-437), the same draws; ``inputs`` is fetched when somebody asks for it, before another dataset
    replaces it on the device, and travels with ``save``."""
    from fokl_gpy_amd.FoKLRoutines import _DeviceInputs
    rng = np.random.default_rng(12)
    n, m = 150_000, 4
    x = rng.uniform(-3.0, 7.0, (n, m)) * np.array([1.0, 1e-3, 250.0, 1.0])
    x[:, 3] += 1e6
    y = np.sin(x[:, 0]) + 0.3 * (x[:, 2] / 250.0) ** 2 + 0.05 * rng.standard_normal(n)
    kw = dict(kernel='Bernoulli Polynomials', UserWarnings=False, ConsoleOutput=False, burnin=30, draws=30)

    def fit(mode):
        monkeypatch.setenv('FOKL_CLEAN', mode)
        model = FoKLRoutines.FoKL(**kw)
        np.random.seed(3)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out = model.fit(x, y, clean=True)
        return model, out

    host, (hb, hm, he) = fit('host')
    assert isinstance(host.__dict__['inputs'], np.ndarray)
    dev, (db, dm, de) = fit('device')
    assert isinstance(dev.__dict__['inputs'], _DeviceInputs) and dev.__dict__['inputs'].shape == (n, m)
    assert dev.minmax == host.minmax
    assert np.array_equal(dm, hm) and np.array_equal(de, he) and np.array_equal(db, hb)
    # another model's dataset arrives on the same device: the first model fetches its inputs before they are replaced
    other, _ = fit('device')
    assert isinstance(dev.__dict__['inputs'], np.ndarray) and isinstance(other.__dict__['inputs'], _DeviceInputs)
    assert np.array_equal(dev.inputs, host.inputs) and dev.inputs.flags.c_contiguous
    # a saved model carries the array
    path = other.save('lazy.fokl', str(tmp_path))
    loaded = FoKLRoutines.load(path)
    assert np.array_equal(loaded.inputs, host.inputs) and np.array_equal(other.inputs, host.inputs)
    assert np.array_equal(x[:, 3] > 0, np.ones(n, dtype=bool))                 # the caller's array is untouched
