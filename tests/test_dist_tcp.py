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


class FakeContext:
    """What dist.RcclComm touches of a _capi.DeviceContext."""

    def __init__(self, behaviour):
        self.behaviour = behaviour

    @staticmethod
    def comm_unique_id():
        return bytes(range(128))

    def comm_init(self, unique_id, rank, world):
        assert bytes(unique_id) == bytes(range(128))
        if self.behaviour == 'fail':
            raise RuntimeError('ncclCommInitRank: unhandled error (stand-in)')
        if self.behaviour == 'hang':
            time.sleep(3600)

    def allgather(self, values, world):
        raise AssertionError("no RCCL collective may run unless RCCL came up on every rank")

    def allreduce_sum(self, values):
        raise AssertionError("no RCCL collective may run unless RCCL came up on every rank")

    def sync(self):
        pass

    def comm_destroy(self):
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
        try:
            comm, kind = dist.bring_up(FakeContext(behaviours[rank]), rank, world, need_rccl, timeout_s=3.0)
            res['kind'] = kind
            if not isinstance(comm, dist.RcclComm):
                res['after'] = comm.allgather([rank + 0.5]).tolist()
                comm.barrier()
            comm.close()
        except RuntimeError as exc:
            res['raised'] = str(exc)
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
