"""
The launch plan of the tile-list Gram kernel (fokl_gram path 2, csrc/fokl_hip.hip: plan_gram) is host arithmetic:
replayed here with numpy -- every tile list executed as the kernel executes it (staged column tiles, the k-steps of a
wavefront team, the slab reduction with the column permutation and the mirror image of the tiles below the diagonal) --
the result must be the block X_r' X_c exactly (integer data), for every shape the search produces and a few it does not.
No device needed.
"""
import numpy as np
import pytest

from fokl_gpy_amd import _capi


def replay(cols, rs, cs, kind=0):
    n = cols.shape[0]
    pl = _capi.gram_plan(rs, cs, kind)
    ks, IT, JT, nci = pl['ks'], pl['i_tiles'], pl['j_tiles'], pl['nci']
    assert pl['rows_per_chunk'] in (32, 64, 128, 256, 512) and pl['ct'] * pl['rows_per_chunk'] // 32 <= 16
    assert 1 <= pl['nt'] <= (4 if kind else 8 if pl['half'].any() else 10) and 1 <= pl['ct'] <= (8 if kind else 16)
    assert ks in (1, 2, 4) and (ks == 1 or pl['nt'] == 1) and pl['waves'] == 4 and (ks == 1 or not kind)
    assert np.array_equal(pl['icols'][:len(rs)], rs)
    X = np.zeros((n, 16 * JT))
    X[:, :nci] = cols[:, pl['icols']]
    slab = np.full((ks, 16 * IT, 16 * JT), np.nan)             # positions nobody computes stay NaN
    rows = np.arange(n)
    written = set()
    for g in range(pl['tiles'].shape[0]):
        st = pl['staged'][g]
        assert np.all(st[pl['ct']:] == -1)
        for w in range(4):
            for k in range(10):
                a, b, oi, oj = pl['tiles'][g, w, k]
                if oi < 0:
                    assert 0 <= a < pl['ct'] and 0 <= b < pl['ct']        # padding tiles still read valid LDS
                    continue
                half_slots = bool(pl['half'].any())                       # entries 0, 1 of every list: half-tile slots
                assert st[a] == oi and st[b] == oj and oj >= oi and (k - 2 if half_slots else k) < pl['nt']
                phase = w % ks
                assert (oi, oj, phase) not in written
                written.add((oi, oj, phase))
                sel = ((rows % 32) // 4) % ks == phase                    # the k-steps this wavefront multiplies
                if pl['half'][g, w, k]:
                    # formed as an 8 x 16 half tile: only where rows 8 .. 15 of the tile are nobody's (they stay NaN here)
                    assert k < 2 and oi == IT - 1 and 1 <= len(rs) % 16 <= 8 and ks == 1 and not kind
                    slab[phase, 16 * oi:16 * oi + 8, 16 * oj:16 * oj + 16] = \
                        X[sel][:, 16 * oi:16 * oi + 8].T @ X[sel][:, 16 * oj:16 * oj + 16]
                    continue
                assert not (half_slots and k < 2)
                slab[phase, 16 * oi:16 * oi + 16, 16 * oj:16 * oj + 16] = \
                    X[sel][:, 16 * oi:16 * oi + 16].T @ X[sel][:, 16 * oj:16 * oj + 16]
    total = slab.sum(axis=0)
    out = np.empty((len(rs), len(cs)))
    for i in range(len(rs)):
        for j in range(len(cs)):
            c, r = int(pl['perm'][j]), i
            if (c >> 4) < (r >> 4):
                r, c = c, r
            out[i, j] = total[r, c]
    return out, pl



def _kinds():
    """Tile-list kinds the library plans: 0 (16x16x4 MFMA, the product) and -- development builds only (make DEV=1) -- 1, the
    retired 4x4x4 form."""
    try:
        _capi.gram_plan(np.array([2], dtype=np.int32), np.array([0, 2, 1], dtype=np.int32), kind=1)
        return (0, 1)
    except _capi.FoklNativeError as exc:
        assert 'development build' in str(exc)
        return (0,)


KINDS = _kinds()

SHAPES = [(1, 1), (2, 3), (8, 10), (16, 16), (17, 33), (28, 38), (56, 58), (56, 66), (65, 131), (70, 150), (56, 176),
          (33, 40), (48, 48), (200, 60), (120, 300)]


@pytest.mark.parametrize('nr,nc', SHAPES)
def test_tile_lists_reproduce_the_block(nr, nc):
    rng = np.random.default_rng(nr * 1000 + nc)
    n = 96
    cols = rng.integers(-3, 4, size=(n, 400)).astype(np.float64)
    # (a) the search's own pattern [ones | model | new | y] with the new columns on the row side, (b) unrelated random
    # lists with partial overlap, (c) lists with repeated slots
    new = rng.permutation(400)[:nr]
    rest = np.setdiff1d(np.arange(400), new)
    search = np.concatenate([rest[:max(nc - nr, 0) // 2], new, rest[200:200 + max(nc - nr, 0)]])[:nc]
    cases = [(new, search), (rng.permutation(400)[:nr], rng.permutation(400)[:nc]),
             (rng.integers(0, 40, nr), rng.integers(0, 40, nc))]
    for rs, cs in cases:
        rs, cs = rs.astype(np.int32), cs.astype(np.int32)
        for kind in KINDS:
            out, _ = replay(cols, rs, cs, kind)
            assert np.array_equal(out, cols[:, rs].T @ cols[:, cs]), kind


def test_symmetric_part_is_computed_once_and_work_is_balanced():
    # 56 new columns against [ones | 118 model columns | the 56 | y]: 4 x 11 tiles, the 6 below the diagonal skipped
    rs = np.arange(100, 156, dtype=np.int32)
    cs = np.concatenate([[0], np.arange(200, 318), rs, [1]]).astype(np.int32)
    pl = _capi.gram_plan(rs, cs)
    real = pl['tiles'][..., 2] >= 0
    assert pl['i_tiles'] == 4 and pl['j_tiles'] == 11 and int(real.sum()) == 38
    per_wave = real.sum(axis=2)
    assert per_wave.max() - per_wave.min() <= 1
    # its last row tile holds 8 of the 56 columns: the 8 tiles of that row sit in the half-tile slots (entries 0 and 1 of
    # the four lists; half the matrix-pipe time each in gram_tiles_dma_kernel), the 30 others follow from entry 2 -- 8 of
    # them per list at most -- and the lists carry 9, 9, 8, 8 tiles' worth of work where plain dealing gives 10, 10, 9, 9
    assert int(pl['half'].sum()) == 8 and np.all(pl['half'][0, :, :2]) and np.all(pl['tiles'][0, :, :2, 2] == 3)
    assert pl['nt'] == 8 and not pl['half'][0, :, 2:].any()
    work = 2 * real.sum(axis=2) - pl['half'].sum(axis=2)
    assert sorted(work[0].tolist()) == [16, 16, 18, 18]
    # the same block for the 4x4x4 kernel: 16 tiles per group at most, lists packed (the kernel skips the MFMAs of
    # padding entries by counting the real ones)
    if 1 in KINDS:
        p4 = _capi.gram_plan(rs, cs, kind=1)
        real4 = p4['tiles'][..., 2] >= 0
        assert int(real4.sum()) == 38 and p4['nt'] <= 4 and p4['ct'] <= 8 and p4['tiles'].shape[0] >= 3
        assert np.all(np.diff(real4.astype(int), axis=2) <= 0)
    # a big block (configs[3]): every group within the kernel's limits, nearly half of the square part skipped
    rs = np.arange(1000, 1560, dtype=np.int32)
    cs = np.concatenate([[0], np.arange(2, 26), rs, [1]]).astype(np.int32)
    pl = _capi.gram_plan(rs, cs)
    real = pl['tiles'][..., 2] >= 0
    full = pl['i_tiles'] * pl['j_tiles']
    assert pl['i_tiles'] == 35 and pl['j_tiles'] == 37 and int(real.sum()) == full - 35 * 34 // 2
    assert pl['ks'] == 1 and pl['rows_per_chunk'] == 32


def test_narrow_blocks_split_the_k_steps_over_wavefronts():
    pl = _capi.gram_plan(np.arange(2, 10, dtype=np.int32), np.concatenate([[0], np.arange(2, 10), [1]]).astype(np.int32))
    assert pl['i_tiles'] == 1 and pl['j_tiles'] == 1 and pl['ks'] == 4 and pl['nt'] == 1
    assert np.all(pl['tiles'][0, :, 0, 2:] == 0)              # all four wavefronts work on the one tile
