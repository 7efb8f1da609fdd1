"""The random stream in two phases (csrc/fokl_stream.cpp): bulk threads produce the MT19937 words, the doubles and the
accept flags of both pairings; one serial walk records tapes as rows of positions; consumers expand rows into numbers.
Everything here is bit for bit: against the one-thread recorder (fokl_noise_tape, itself pinned against numpy in
tests/test_sampler_host.py), against numpy directly, between the AVX-512 walk (positions first, accept tests eight at a
time), the scalar walk and the portable build, across rewinds and segment boundaries."""
import math
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from fokl_gpy_amd import _capi


def _state(seed, pos=None, cached=None):
    np.random.seed(seed)
    st = np.random.get_state()
    return ('MT19937', st[1], st[2] if pos is None else pos, 0 if cached is None else 1,
            0.0 if cached is None else cached)


def _shapes(p1, n=1e6, a=4.0, atau=4.0):
    return a + 1 + n / 2 + p1 / 2, atau + (p1 - 1) / 2                # FR:1508, FR:1510


def _old_tapes(state, sizes, draws, shapes=_shapes):
    stream = _capi.LegacyStream(state)
    return [_capi.record_noise_tape(_capi.NoiseTape(p1, draws), *shapes(p1), stream) for p1 in sizes], stream


def _new_tapes(state, sizes, draws, shapes=_shapes, bulk_threads=2, helpers=0):
    stream = _capi.LegacyStream(state)
    eng = _capi.StreamEngine(stream, bulk_threads)
    if helpers:
        _capi._check(_capi.load().fokl_stream_set_helpers(eng._h, helpers, None))
    tapes = []
    for p1 in sizes:
        tape = _capi.NoiseTape(p1, draws)
        hold = eng.walk(tape, *shapes(p1))
        eng.expand(tape, *shapes(p1))
        eng.release(hold)
        tapes.append(tape)
    eng.close()
    return tapes, stream


def _assert_same_tape(new, old):
    p1 = old.p1
    assert np.array_equal(new.lead, old.lead)
    assert np.array_equal(new.gam_sig, old.gam_sig) and np.array_equal(new.gam_tau, old.gam_tau)
    assert np.array_equal(new.normals, old.normals)
    for k in range(old.draws):
        n = (p1 - old.lead[k]) // 2
        assert np.array_equal(new.pair_r2[k, :n], old.pair_r2[k, :n])


def _same_state(a, b):
    return np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize('seed,pos', [(0, None), (1, 1), (2, 311), (3, 622), (4, 623), (5, 624), (6, 0), (7, 2)])
def test_walk_and_expansion_equal_the_one_thread_recorder(seed, pos):
    """Tapes of many sizes back to back from even, odd and block-boundary word positions (odd ones: every double
    straddles two 64-bit words and, at block ends, two MT19937 blocks): rows expanded == fokl_noise_tape's arrays, and
    the stream ends on the same numpy state."""
    sizes = [1, 2, 3, 7, 8, 60, 61, 33, 128, 129]
    state = _state(seed, pos)
    old, so = _old_tapes(state, sizes, 150)
    new, sn = _new_tapes(state, sizes, 150)
    for a, b in zip(new, old):
        _assert_same_tape(a, b)
    assert _same_state(sn.as_numpy_state(), so.as_numpy_state())


@pytest.mark.parametrize('helpers', [1, 2, 3])
def test_walk_with_helper_threads_records_the_same_rows(helpers):
    """Round 6: the walking thread only counts accepted attempts and hands blocks of 128 iterations to helper threads
    (positions, rows, accept tests; the walker takes blocks itself while it would only wait).  Rows, expanded numbers and the
    stream's end state are the one-thread walk's bit for bit -- with frequent rejected gamma draws (small shapes: a block that
    is not good takes everything issued behind it with it) as well as with the fit's shapes, several times over (the threads
    meet differently every time)."""
    for shapes in (_shapes, lambda p: (1.2, 1.7)):
        sizes = [70, 7, 300, 2, 33, 150]
        state = _state(9, pos=1)
        want, sw = _new_tapes(state, sizes, 900, shapes)
        for _ in range(3):
            got, sg = _new_tapes(state, sizes, 900, shapes, helpers=helpers)
            for a, b in zip(got, want):
                assert np.array_equal(a.rows, b.rows)
                _assert_same_tape(a, b)
            assert _same_state(sg.as_numpy_state(), sw.as_numpy_state())


def test_bulk_threads_placed_on_given_cpus_and_wide_tapes_far_ahead():
    """Round 6: fokl_stream_place_bulk pins the producers (bad arguments are refused); a walk through models of hundreds of
    columns -- production bound: a segment per microsecond of walking, the producers' distance growing with the walker's waits,
    segments taken before the token -- records the tapes of the plain walk."""
    import os
    lib = _capi.load()
    state = _state(21, pos=5)
    sizes = [585, 3, 585, 440, 585]
    want, sw = _new_tapes(state, sizes, 120, bulk_threads=1)
    stream = _capi.LegacyStream(state)
    eng = _capi.StreamEngine(stream, 3)
    cpus = np.ascontiguousarray(sorted(os.sched_getaffinity(0))[:2], dtype=np.int32)
    _capi._check(lib.fokl_stream_place_bulk(eng._h, _capi._ptr(cpus), cpus.shape[0]))
    assert lib.fokl_stream_place_bulk(eng._h, None, 0) != 0
    got = []
    for p1 in sizes:
        tape = _capi.NoiseTape(p1, 120)
        hold = eng.walk(tape, *_shapes(p1))
        eng.expand(tape, *_shapes(p1))
        eng.release(hold)
        got.append(tape)
    eng.close()
    for a, b in zip(got, want):
        assert np.array_equal(a.rows, b.rows)
        _assert_same_tape(a, b)
    assert _same_state(stream.as_numpy_state(), sw.as_numpy_state())


def test_a_cached_normal_handed_over_with_the_state_opens_the_first_row():
    state = _state(11, pos=77, cached=-0.4321)
    old, so = _old_tapes(state, [5, 4], 40)
    new, sn = _new_tapes(state, [5, 4], 40)
    assert old[0].lead[0] == 1 and old[0].normals[0, 0] == -0.4321
    for a, b in zip(new, old):
        _assert_same_tape(a, b)
    assert _same_state(sn.as_numpy_state(), so.as_numpy_state())
    # nothing walked: the state goes back untouched, cached value included
    stream = _capi.LegacyStream(state)
    eng = _capi.StreamEngine(stream, 1)
    eng.close()
    assert _same_state(stream.as_numpy_state(), state)


def test_expanded_rows_are_numpys_draws():
    """Straight against numpy: normals completed with libm's log are np.random.normal's, the gammas np.random.gamma's."""
    p1, draws = 9, 300
    astar, atau_star = 7.25, 3.5
    np.random.seed(2024)
    want_n, want_s, want_t = [], [], []
    for _ in range(draws):
        want_n.append(np.random.normal(0, 1, size=(p1, 1)).ravel())
        want_s.append(np.random.gamma(astar, 1.0))
        want_t.append(np.random.gamma(atau_star, 1.0))
    end = np.random.get_state()
    tapes, stream = _new_tapes(_state(2024), [p1], draws, shapes=lambda p: (astar, atau_star))
    tape = tapes[0]
    raw, r2 = tape.normals.copy(), tape.pair_r2
    for k in range(draws):
        ld = tape.lead[k]
        n = (p1 - ld) // 2
        f = np.array([math.sqrt(-2.0 * math.log(float(v)) / float(v)) for v in r2[k, :n]])       # libm, as numpy's C
        raw[k, ld:ld + 2 * n] *= np.repeat(f, 2)
    assert np.array_equal(raw, np.array(want_n))
    assert np.array_equal(tape.gam_sig, np.array(want_s)) and np.array_equal(tape.gam_tau, np.array(want_t))
    assert _same_state(stream.as_numpy_state(), end)


@pytest.mark.parametrize('astar,atau_star', [(1.02, 1.3), (0.4, 2.0), (1.0, 0.0), (3.0, 0.999), (1.5, 1.0001)])
def test_small_shapes_and_frequent_rejections(astar, atau_star):
    """Shapes near 1 reject often (the AVX-512 walk rolls those iterations back and redoes them draw by draw); shapes
    <= 1 take numpy's other branches (the walker stores the variate itself)."""
    state = _state(31, pos=5)
    shapes = lambda p: (astar, atau_star)
    old, so = _old_tapes(state, [6, 7, 1], 400, shapes)
    new, sn = _new_tapes(state, [6, 7, 1], 400, shapes)
    for a, b in zip(new, old):
        _assert_same_tape(a, b)
    assert _same_state(sn.as_numpy_state(), so.as_numpy_state())


def test_long_walk_over_many_segments_with_rewinds():
    """More than a million doubles: some twenty 79 872-double segments made by three bulk threads, recycled behind the walker;
    a rewind (what an aborted tentative tape does) replays the same rows; the state in between is numpy's."""
    state = _state(77, pos=3)
    sizes = [70, 71, 585, 12, 586, 70]
    draws = 700
    old, so = _old_tapes(state, sizes, draws)
    stream = _capi.LegacyStream(state)
    eng = _capi.StreamEngine(stream, 3)
    ref = _capi.LegacyStream(state)
    for i, p1 in enumerate(sizes):
        first = _capi.NoiseTape(p1, draws)
        at = eng.tell()
        hold = eng.walk(first, *_shapes(p1))
        if i % 2 == 0:                                          # send it back, walk something else, send that back too
            eng.seek(at)
            other = _capi.NoiseTape(p1 + 3, draws // 2)
            h2 = eng.walk(other, *_shapes(p1 + 3))
            eng.release(h2)
            eng.seek(at)
            again = _capi.NoiseTape(p1, draws)
            h3 = eng.walk(again, *_shapes(p1))
            assert np.array_equal(again.rows, first.rows)
            eng.release(h3)
        eng.expand(first, *_shapes(p1))
        eng.release(hold)
        _assert_same_tape(first, old[i])
        _capi.record_noise_tape(_capi.NoiseTape(p1, draws), *_shapes(p1), ref)
        assert _same_state(eng.numpy_state(), ref.as_numpy_state())
    stats = eng.stats()
    eng.close()
    assert _same_state(stream.as_numpy_state(), so.as_numpy_state())
    assert stats['segments'] >= 15
    # the bounds decide nearly every accept test; the exact expressions are the exception
    assert stats['gamma_attempts_exact'] < 0.02 * stats['gamma_attempts']


def test_rows_expand_in_pieces_and_out_of_order():
    state = _state(5)
    p1, draws = 37, 256
    stream = _capi.LegacyStream(state)
    eng = _capi.StreamEngine(stream, 2)
    whole, pieces = _capi.NoiseTape(p1, draws), _capi.NoiseTape(p1, draws)
    hold = eng.walk(whole, *_shapes(p1))
    pieces.rows[:] = whole.rows
    pieces.gam_sig[:], pieces.gam_tau[:] = whole.gam_sig, whole.gam_tau
    eng.expand(whole, *_shapes(p1))
    for first in (192, 0, 64, 128):
        eng.expand(pieces, *_shapes(p1), first=first, last=first + 64)
    eng.release(hold)
    eng.close()
    _assert_same_tape(pieces, whole)


def test_the_bounds_rest_on_a_log_that_is_accurate_enough():
    """The walker accepts a gamma draw without forming its normal when bounds built on fast_ln decide the test; their
    slack (1e-4 on ln r2) must dwarf the approximation's error."""
    assert _capi.load().fokl_stream_fast_ln_error(400000) < 1e-6


def test_walks_of_the_three_builds_agree():
    """AVX-512 walk (positions first, accept tests eight at a time, roll-backs), the scalar walk on the same machine code
    (FOKL_STREAM_SCALAR_WALK=1) and the portable build (FOKL_SAMPLER_ISA=base) record identical rows; each in a fresh
    process, the choice is made once per process."""
    code = textwrap.dedent('''
        import sys, hashlib
        import numpy as np
        sys.path.insert(0, %r)
        from tests.test_stream_engine import _new_tapes, _state, _shapes
        h = hashlib.sha256()
        for shapes in (_shapes, lambda p: (1.2, 1.7)):
            tapes, stream = _new_tapes(_state(9, pos=1), [70, 7, 300, 2], 500, shapes)
            for t in tapes:
                for part in (t.rows, t.normals, t.gam_sig, t.gam_tau, t.lead):
                    h.update(np.ascontiguousarray(part).tobytes())
            h.update(stream.key.tobytes())
        print(h.hexdigest())
    ''') % (str(__import__('pathlib').Path(__file__).resolve().parents[1]),)
    import os
    seen = {}
    for name, env in (('wide', {}), ('scalar', {'FOKL_STREAM_SCALAR_WALK': '1'}), ('base', {'FOKL_SAMPLER_ISA': 'base'})):
        out = subprocess.run([sys.executable, '-c', code], env={**os.environ, **env}, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr
        seen[name] = out.stdout.strip()
    assert len(set(seen.values())) == 1, seen


def test_bad_arguments_are_rejected():
    stream = _capi.LegacyStream(_state(1))
    eng = _capi.StreamEngine(stream, 1)
    tape = _capi.NoiseTape(3, 8)
    with pytest.raises(_capi.FoklNativeError):
        eng.walk(tape, -1.0, 2.0)
    assert tape.progress[0] == -1
    with pytest.raises(_capi.FoklNativeError):
        eng.release(12345)                                      # no such hold
    eng.close()
    with pytest.raises(_capi.FoklNativeError):
        bad = _capi.LegacyStream(_state(1))
        bad.pos = __import__('ctypes').c_int32(700)
        _capi.StreamEngine(bad, 1)
