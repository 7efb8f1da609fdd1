"""
G3 on the device (include/fokl_hip.h: fokl_dchain_*): the finishing of the polar normals and the Gibbs recursion of
FR:1519-1548 on the GPU against the host chain (fokl_gibbs_chain_from_tape, itself pinned bit for bit to the
reference's loop and numpy's stream in tests/test_sampler_host.py) on the same noise tapes.

Tolerances: with FOKL_DCHAIN_RECURSION=exact a tape whose normals were finished on the host goes through the device
recursion BIT FOR BIT (same IEEE operations in the same order, the quadratic forms summed in the host's order); with the
finishing on the device the normals differ from libm's in the last bit of log(), and the default recursion folds its
reciprocals (three long operations on the critical path instead of nine), each fold a rounding or two away from the
host's: both reach the draws at the 1e-16 .. 1e-15 level.  Round 6's recursion forms bstar as
b + (dtd - sum d qty^2 + sigma^2 sum v^2) / 2 -- two of its three sums off the critical path -- where the host forms
b + (w'Lw - 2 w'qty + dtd + w'w / tau^2) / 2: the same cancellation against dtd reached through different roundings, a
digit's worth on models whose fit is nearly exact.  Bounded here by TOL = 1e-12 of the column scale over the full 2000
iterations of a fit's chain (the reference's draws are held to 1e-9, tests/test_config_goldens.py).
"""
import os

import numpy as np
import pytest

from fokl_gpy_amd import _capi

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope='module')
def engine():
    """The default engine: the recursion with its reciprocals folded (three long operations on the critical path)."""
    eng = _capi.DeviceChainEngine(int(os.environ.get('FOKL_DEVICE', '0')), slots=8)
    yield eng
    eng.close()


@pytest.fixture(scope='module')
def exact_engine():
    """FOKL_DCHAIN_RECURSION=exact: the host chain's operations in the host chain's order."""
    saved = os.environ.get('FOKL_DCHAIN_RECURSION')
    os.environ['FOKL_DCHAIN_RECURSION'] = 'exact'
    try:
        eng = _capi.DeviceChainEngine(int(os.environ.get('FOKL_DEVICE', '0')), slots=8)
    finally:
        if saved is None:
            del os.environ['FOKL_DCHAIN_RECURSION']
        else:
            os.environ['FOKL_DCHAIN_RECURSION'] = saved
    yield eng
    eng.close()


def model(p1, rng):
    lamb = np.sort(rng.random(p1) * 1e5 + 1e-2)
    lamb[0] = 3e-3
    qty = rng.standard_normal(p1) * np.sqrt(lamb) * 3
    return lamb, qty


def host_tape(p1, draws, seed, exact=True):
    np.random.seed(seed)
    stream = _capi.LegacyStream()
    return _capi.noise_tape(p1, draws, 500.0 + p1 / 2, 4 + (p1 - 1) / 2, stream)


@pytest.mark.parametrize('which', ['fast', 'exact'])
@pytest.mark.parametrize('p1', [1, 2, 7, 17, 60, 64, 65, 129, 200, 586])
def test_device_chain_equals_the_host_chain_on_the_same_tape(engine, exact_engine, which, p1, monkeypatch):
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    engine = engine if which == 'fast' else exact_engine
    rng = np.random.default_rng(p1)
    draws = 300 if p1 > 200 else 2000
    lamb, qty = model(p1, rng)
    tape = host_tape(p1, draws, 10 + p1)
    args = (lamb, qty, 900.0, 2.0, 5e5, 0.3, 0.9)
    want, flag = _capi.gibbs_chain_from_tape(*args, tape)
    half0 = draws // 2
    w, mean_w, negative = _capi.gibbs_chain_device(engine, *args, tape, stat_first=half0)
    assert not negative and not flag
    scale = np.max(np.abs(want), axis=0)
    assert np.max(np.abs(w - want) / scale) < TOL
    assert np.max(np.abs(mean_w - want[half0:].mean(axis=0)) / scale) < TOL


@pytest.mark.parametrize('p1', [3, 60, 150])
def test_device_recursion_is_bitwise_the_host_recursion_on_finished_normals(exact_engine, p1, monkeypatch):
    """Finishing done by the host (exact log): what is left for the device is the recursion -- division, square root,
    products and the three sums in the host's order -- and the draws come out bit for bit."""
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    rng = np.random.default_rng(100 + p1)
    draws = 500
    lamb, qty = model(p1, rng)
    tape = host_tape(p1, draws, 77 + p1)
    args = (lamb, qty, 40.0, 1.5, 2e4, 0.2, 1.1)
    _capi.finish_tape_blocks(tape)
    want, _ = _capi.gibbs_chain_from_finished_tape(*args, tape)
    tape.finishing_requested = True
    job = exact_engine.submit(*args, tape, stat_first=0, follow=False)
    try:
        job.wait()
        w = job.fetch_w()
        sig, tau = job.last_state
    finally:
        job.release()
    assert np.array_equal(w, want)
    assert np.isfinite(sig) and np.isfinite(tau)


def test_negative_bstar_is_flagged_and_goes_nan_like_the_host(engine, monkeypatch):
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    rng = np.random.default_rng(5)
    lamb, qty = model(6, rng)
    tape = host_tape(6, 50, 3)
    args = (lamb, qty, -1e9, 2.0, 10.0, 0.3, 0.9)                     # b far below zero: bstar < 0 at once
    want, flag = _capi.gibbs_chain_from_tape(*args, tape)
    w, mean_w, negative = _capi.gibbs_chain_device(engine, *args, tape)
    assert flag and negative
    assert np.array_equal(np.isnan(w), np.isnan(want)) and np.isnan(w[1:]).all()
    # (the first row is formed before anything goes NaN: the fast recursion's Newton-refined reciprocal square root is within
    # a rounding or two of the host's 1 / x and sqrt)
    assert np.allclose(w[0], want[0], rtol=1e-14, atol=0)


def test_chains_follow_tapes_that_are_still_on_record_and_slots_are_recycled(engine):
    """Jobs are submitted while the pool's noise thread is still recording the tapes (the dispatcher waits for each
    tape's progress flag); more jobs than slots go through as slots are released; a full engine says so."""
    rng = np.random.default_rng(9)
    np.random.seed(4)
    s_pool, s_ref = _capi.LegacyStream(), _capi.LegacyStream()
    pool = _capi.HostPool(s_pool, chain_threads=1, finish_threads=0, spectral_threads=0)
    draws, jobs = 400, []
    sizes = [40, 61, 8, 130, 60, 60, 59, 58, 57, 12, 300, 5]
    try:
        for p in sizes:
            lamb, qty = model(p, rng)
            tape = _capi.NoiseTape(p, draws)
            noise = pool.submit_noise(tape, 500.0 + p / 2, 4 + (p - 1) / 2)
            args = (lamb, qty, 900.0, 2.0, 5e5, 0.3, 0.9)
            while True:
                try:
                    job = engine.submit(*args, tape, stat_first=draws // 2)
                    break
                except _capi.FoklNativeError as exc:                    # all 8 slots alive: release the oldest
                    assert exc.code == -3
                    old = jobs.pop(0)
                    check(old, draws, s_ref)
            jobs.append((job, noise, args, p))
        while jobs:
            check(jobs.pop(0), draws, s_ref)
    finally:
        pool.close()
    assert engine.stats()['issued'] >= len(sizes)


def check(entry, draws, s_ref):
    job, noise, args, p = entry
    mean_w, flag = job.wait()
    noise.wait()
    w = job.fetch_w()
    job.release()
    want = _capi.gibbs_chain(args[0], args[1], 500.0 + p / 2, 4 + (p - 1) / 2, *args[2:], draws, s_ref)
    scale = np.max(np.abs(want), axis=0)
    assert flag[0] == 0 and np.max(np.abs(w - want) / scale) < TOL
    assert np.max(np.abs(mean_w - want[draws // 2:].mean(axis=0)) / scale) < TOL


@pytest.mark.parametrize('draws', [1, 2, 7, 8, 9, 17, 129])
def test_short_chains_and_ring_boundaries(engine, exact_engine, draws, monkeypatch):
    """Fewer iterations than the prefetch ring holds, one more than it holds, one more than a staged block of gammas:
    the clamped row indices of the ring and the gamma blocks must not leak into the results."""
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    rng = np.random.default_rng(draws)
    for p1 in (3, 70):
        lamb, qty = model(p1, rng)
        tape = host_tape(p1, draws, 5 * draws + p1)
        args = (lamb, qty, 900.0, 2.0, 5e5, 0.3, 0.9)
        want, _ = _capi.gibbs_chain_from_tape(*args, tape)
        for eng in (engine, exact_engine):
            w, mean_w, negative = _capi.gibbs_chain_device(eng, *args, tape, stat_first=draws // 2)
            scale = np.max(np.abs(want), axis=0)
            assert not negative and np.max(np.abs(w - want) / scale) < TOL
            assert np.max(np.abs(mean_w - want[draws // 2:].mean(axis=0)) / scale) < TOL


def test_largest_model_and_beyond(engine, monkeypatch):
    """768 columns is what one wavefront holds (12 eigen-directions per lane); beyond that the engine says so and the
    search keeps such chains on the host threads (engine.ForwardSelection.device_chain_columns)."""
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    rng = np.random.default_rng(768)
    lamb, qty = model(768, rng)
    tape = host_tape(768, 40, 768)
    args = (lamb, qty, 900.0, 2.0, 5e5, 0.3, 0.9)
    want, _ = _capi.gibbs_chain_from_tape(*args, tape)
    w, _, negative = _capi.gibbs_chain_device(engine, *args, tape)
    assert not negative and np.max(np.abs(w - want) / np.max(np.abs(want), axis=0)) < TOL
    lamb, qty = model(769, rng)
    with pytest.raises(_capi.FoklNativeError) as err:
        engine.submit(lamb, qty, 900.0, 2.0, 5e5, 0.3, 0.9, host_tape(769, 10, 1), follow=False)
    assert err.value.code == -2 and '768' in str(err.value)


def test_tapes_in_ordinary_memory_are_staged(engine, monkeypatch):
    """A tape in page-locked memory (fokl_host_alloc) is read by the device in place; one in ordinary memory goes through
    copy calls and a device staging buffer: same draws either way."""
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    rng = np.random.default_rng(11)
    p1, draws = 45, 300
    lamb, qty = model(p1, rng)
    args = (lamb, qty, 900.0, 2.0, 5e5, 0.3, 0.9)
    np.random.seed(5)
    s1, s2 = _capi.LegacyStream(), _capi.LegacyStream()
    plain = _capi.noise_tape(p1, draws, 500.0, 26.0, s1)
    need = _capi.NoiseTape.doubles_needed(p1, draws) + 8
    pinned = _capi.record_noise_tape(_capi.NoiseTape(p1, draws, _capi.pinned_empty(need)), 500.0, 26.0, s2)
    before = engine.stats()['staged']
    w_plain, _, _ = _capi.gibbs_chain_device(engine, *args, plain)
    assert engine.stats()['staged'] == before + 1
    w_pinned, _, _ = _capi.gibbs_chain_device(engine, *args, pinned)
    assert engine.stats()['staged'] == before + 1
    assert np.array_equal(w_plain, w_pinned)


@pytest.mark.parametrize('p1,pos', [(1, 0), (2, 1), (7, 623), (60, 2), (61, 311), (129, 624), (586, 5)])
def test_rows_expanded_on_the_device_equal_the_tape_expanded_on_the_host(p1, pos, monkeypatch):
    """Round 4: a tape as 32-byte rows (fokl_stream_walk).  The engine regenerates the stream's segments from the raw
    pre-states the host's bulk threads leave in its ring (MT19937 recurrence, tempering, numpy's doubles: exact) and
    expands the rows there -- accepted attempts re-decided from x1^2 + x2^2 with the host's roundings -- so the chain it
    runs is the host chain on the host-expanded tape up to log()."""
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    eng = _capi.DeviceChainEngine(int(os.environ.get('FOKL_DEVICE', '0')), slots=8)
    try:
        np.random.seed(100 + p1)
        st = np.random.get_state()
        stream = _capi.LegacyStream(('MT19937', st[1], pos, 0, 0.0))
        se = _capi.StreamEngine(stream, 2, prestates=eng.prestate_ring())
        eng.bind(se._h)
        rng = np.random.default_rng(p1)
        draws = 300 if p1 > 200 else 2000
        astar, atau_star = 5e5 + p1 / 2, 4 + (p1 - 1) / 2
        args = (900.0, 2.0, 5e5, 0.3, 0.9)
        jobs, wants, holds = [], [], []
        for rep in range(3):                                  # three tapes back to back: later ones start mid-segment
            lamb, qty = model(p1, rng)
            tape = _capi.NoiseTape(p1, draws, raw=_capi.pinned_empty(_capi.NoiseTape.doubles_needed(p1, draws)))
            first = se.tell().position
            holds.append(se.walk(tape, astar, atau_star))
            span = np.array([holds[-1], se.tell().position], dtype=np.uint64)
            assert span[0] <= first
            jobs.append(eng.submit_rows(lamb, qty, *args, astar, atau_star, tape, span, stat_first=draws // 2))
            host = _capi.NoiseTape(p1, draws)
            host.rows[:] = tape.rows
            se.expand(host, astar, atau_star)
            host.progress[0] = draws
            wants.append(_capi.gibbs_chain_from_tape(lamb, qty, *args, host)[0])
        for job, want in zip(jobs, wants):
            mean_w, flag = job.wait()
            w = job.fetch_w()
            scale = np.max(np.abs(want), axis=0)
            assert not flag[0]
            assert np.max(np.abs(w - want) / scale) < TOL
            assert np.max(np.abs(mean_w - want[draws // 2:].mean(axis=0)) / scale) < TOL
            job.release()
        assert eng.stream_stats()['rows_jobs'] == 3 and eng.stream_stats()['segments_made'] >= 1
        for hold in holds:
            se.release(hold)
        eng.bind(None)
        se.close()
    finally:
        eng.close()


@pytest.mark.parametrize('p1', [586, 768])
def test_sixteen_wide_chains_at_once_at_the_full_chain_length(p1, monkeypatch):
    """the size classes above four elements per lane (T = 12: up to 768 columns) and tapes in page-locked buffers
    of exactly the size they need had no clean batched run on record (tools/chain_device_probe.py died at 586 columns).
    Sixteen chains of 2000 iterations in flight together, from tapes recorded into exact-size page-locked buffers (the
    one-thread recorder and its whole-vector stores included) and from rows expanded on the device: every one equals the
    host chain on the same tape."""
    monkeypatch.setenv('FOKL_FINISH_LOG', 'exact')
    draws, chains = 2000, 16
    eng = _capi.DeviceChainEngine(int(os.environ.get('FOKL_DEVICE', '0')), slots=2 * chains)
    try:
        rng = np.random.default_rng(p1)
        args = (900.0, 2.0, 5e5, 0.3, 0.9)
        astar, atau_star = 5e5 + p1 / 2, 4 + (p1 - 1) / 2
        np.random.seed(p1)
        old_stream = _capi.LegacyStream()
        se = _capi.StreamEngine(_capi.LegacyStream(), 3, prestates=eng.prestate_ring())
        eng.bind(se._h)
        jobs, holds = [], []
        for c in range(chains):
            lamb, qty = model(p1, rng)
            raw = _capi.pinned_empty(_capi.NoiseTape.doubles_needed(p1, draws))
            tape = _capi.NoiseTape(p1, draws, raw=raw)
            if c % 2 == 0:
                # the round-3 path: the one-thread recorder into the page-locked buffer, gathered over the bus
                _capi.record_noise_tape(tape, astar, atau_star, old_stream)
                job = eng.submit(lamb, qty, *args, tape, stat_first=draws // 2, follow=False)
                want = _capi.gibbs_chain_from_tape(lamb, qty, *args, tape)[0]
            else:
                # the round-4 path: rows, expanded on the device
                holds.append(se.walk(tape, astar, atau_star))
                span = np.array([holds[-1], se.tell().position], dtype=np.uint64)
                job = eng.submit_rows(lamb, qty, *args, astar, atau_star, tape, span, stat_first=draws // 2)
                host = _capi.NoiseTape(p1, draws)
                host.rows[:] = tape.rows
                se.expand(host, astar, atau_star)
                host.progress[0] = draws
                want = _capi.gibbs_chain_from_tape(lamb, qty, *args, host)[0]
            jobs.append((job, want))
        for job, want in jobs:
            mean_w, flag = job.wait()
            w = job.fetch_w()
            scale = np.max(np.abs(want), axis=0)
            assert not flag[0]
            assert np.max(np.abs(w - want) / scale) < TOL
            assert np.max(np.abs(mean_w - want[draws // 2:].mean(axis=0)) / scale) < TOL
            job.release()
        for hold in holds:
            se.release(hold)
        eng.bind(None)
        se.close()
    finally:
        eng.close()
