"""
The TCP control plane of a multi-rank launch (dist.TcpComm) and the guarded RCCL bring-up (dist.bring_up): three
processes on the CPU, a stand-in device context whose RCCL initialisation succeeds, fails or hangs per rank.  What must
hold: every rank ends up with the SAME kind of communicator, a launch of independent fits survives without RCCL, a launch
that needs RCCL on its data path fails on every rank instead of hanging.
"""
import multiprocessing as mp
import os
import socket
import time

import numpy as np

from fokl_gpy_amd import dist


class FakePending:
    """What dist.bring_up touches of an RcclComm that has initialised but is not attached to a context yet."""

    def __init__(self, ctx, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world

    def attach(self, ctx):
        assert ctx is self.ctx
        ctx.state.append('attached')
        return FakeRccl(ctx)

    def drop(self):
        self.ctx.state.append('dropped')


class FakeRccl(dist.RcclComm):
    def __init__(self, ctx):
        self.ctx = ctx

    def close(self):
        self.ctx.state.append('closed')


class FakeContext:
    """Stand-in for a _capi.DeviceContext whose RCCL initialisation succeeds, fails, hangs or comes back late."""
    device = 0

    def __init__(self, behaviour, rank, world):
        self.behaviour, self.rank, self.world = behaviour, rank, world
        self.state = []                         # what was done to the context, in order

    def rccl_initialise(self):
        # runs on bring_up's helper thread: must not touch the context (self.state) -- only attach / drop may
        if self.behaviour == 'fail':
            raise RuntimeError('ncclCommInitRank: unhandled error (stand-in)')
        if self.behaviour == 'hang':
            time.sleep(3600)
        if self.behaviour == 'late':
            time.sleep(5.0)                     # beyond the deadline of 3 s: comes back when nobody wants it any more
        return FakePending(self, self.rank, self.world)

    def allgather(self, values, world):
        raise AssertionError("no RCCL collective may run unless RCCL came up on every rank")

    def allreduce_sum(self, values):
        raise AssertionError("no RCCL collective may run unless RCCL came up on every rank")

    def sync(self):
        pass


def _worker(rank, world, port, behaviours, need_rccl, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        tcp = dist.TcpComm(rank, world, tag='fokl_test_tcp')
        g = tcp.allgather([rank, 10.0 * rank])
        s = tcp.allreduce_sum(np.array([[1.0, rank], [2.0, 3.0]]))
        tcp.barrier()
        tcp.close()
        res = dict(gather=g.tolist(), reduce=s.tolist())
        fake = FakeContext(behaviours[rank], rank, world)
        try:
            comm, kind = dist.bring_up(fake, rank, world, need_rccl, timeout_s=3.0)
            res['kind'] = kind
            if not isinstance(comm, dist.RcclComm):
                res['after'] = comm.allgather([rank + 0.5]).tolist()
                comm.barrier()
            else:
                # the control plane stays up beside RCCL: barriers and timing gathers of a launch need no device collective
                res['control'] = comm.control.allgather([rank + 0.25]).tolist()
                comm.control.barrier()
                comm.control.close()
            comm.close()
        except RuntimeError as exc:
            res['raised'] = str(exc)
        if behaviours[rank] == 'late':
            time.sleep(4.0)                     # the abandoned helper thread has returned by now
        res['context'] = list(fake.state)
        out.put((rank, res))
    except BaseException as exc:                                  # noqa: BLE001
        out.put((rank, dict(crash=f'{type(exc).__name__}: {exc}')))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(behaviours, need_rccl):
    ctx = mp.get_context('fork')
    out = ctx.Queue()
    world, port = len(behaviours), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, behaviours, need_rccl, out), daemon=True)
             for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(10)
        if p.is_alive():
            p.terminate()
    return [got[r] for r in range(world)]


def test_control_plane_collectives_and_fallback_when_rccl_fails_on_one_rank():
    res = _launch(['ok', 'fail', 'ok'], need_rccl=False)
    for r, one in enumerate(res):
        assert 'crash' not in one, one
        assert one['gather'] == [[0.0, 0.0], [1.0, 10.0], [2.0, 20.0]]
        assert one['reduce'] == [[3.0, 3.0], [6.0, 9.0]]
        assert one['kind'].startswith('TCP control plane only')    # the same decision on every rank
        assert one['after'] == [[0.5], [1.5], [2.5]]


def test_a_hanging_rccl_initialisation_is_survived_by_independent_fits():
    res = _launch(['ok', 'hang'], need_rccl=False)
    assert all('crash' not in one and one['kind'].startswith('TCP control plane only') for one in res), res
    assert 'did not return' in res[1]['kind']


def test_modes_that_need_rccl_fail_everywhere_instead_of_hanging():
    res = _launch(['fail', 'ok'], need_rccl=True)
    assert all('raised' in one and 'did not come up' in one['raised'] for one in res), res


def test_rccl_is_used_when_it_comes_up_everywhere():
    res = _launch(['ok', 'ok'], need_rccl=True)
    assert all(one.get('kind') == 'RCCL' for one in res), res
    assert all(one.get('control') == [[0.25], [1.25]] for one in res), res


def test_a_communicator_that_is_not_used_everywhere_is_dropped_and_a_late_one_never_reaches_the_context():
    """Rank 0 initialises in time, rank 1 only after the deadline: nobody may keep a communicator attached (rank 0 drops
    its own), and what rank 1's abandoned helper thread returns later is never attached to the context the main thread
    has gone on to use (round-2 review: the late thread wrote ctx->comm under the running fit)."""
    res = _launch(['ok', 'late'], need_rccl=False)
    assert all('crash' not in one and one['kind'].startswith('TCP control plane only') for one in res), res
    assert res[0]['context'] == ['dropped'], res[0]
    assert res[1]['context'] == [], res[1]
    assert 'did not return' in res[1]['kind']


def test_attach_happens_only_when_every_rank_is_up():
    res = _launch(['ok', 'ok', 'ok'], need_rccl=True)
    assert all(one.get('kind') == 'RCCL' and one['context'] == ['attached', 'closed'] for one in res), res
    res = _launch(['ok', 'fail', 'ok'], need_rccl=False)
    assert [one['context'] for one in res] == [['dropped'], [], ['dropped']], res


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): the process starts its N ranks itself --
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them, no torch involved -- and the ranks
    meet over dist.TcpComm.  --launch-check stops there (no GPU needed); a launcher that started the wrong number of ranks
    is still refused."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE',
                                                             'MASTER_PORT', 'TORCHELASTIC_RUN_ID')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--launch-check'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line == {'launch_check': [0, 1, 2, 3], 'world': 4, 'local_world': 4}
    wrong = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--launch-check'],
                           env=dict(env, WORLD_SIZE='2', RANK='0'), capture_output=True, text=True, timeout=300)
    assert wrong.returncode == 2 and 'WORLD_SIZE=2' in wrong.stderr
