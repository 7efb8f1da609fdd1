"""
Multi-GPU plumbing: one process per GPU, RCCL over xGMI through the C ABI (``fokl_comm_*``).

The forward-selection path has exactly two exchange steps, both tiny:
  * all-gather of per-candidate BIC values when kill-test proposals (or whole independent fits) are sharded
    over ranks;
  * all-reduce(sum) of Gram blocks / residual moments when *rows* are sharded over ranks.

``RcclComm`` drives them on the GPU; ``GlooComm`` offers the same interface over ``torch.distributed``'s gloo
backend so that the N > 1 host logic is testable on CPU (tests/test_dist_gloo.py).  torch is used for
rendezvous / CPU testing only -- never on the device path.
"""
import os

import numpy as np


def env_rank_world():
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    return rank, world, local


class SingleComm:
    """World of one: every collective is the identity."""
    rank, world = 0, 1

    def allgather(self, values):
        return np.asarray(values, dtype=np.float64)[None, :].copy()

    def allreduce_sum(self, values):
        return np.array(values, dtype=np.float64, copy=True)

    def barrier(self):
        pass

    def close(self):
        pass


def _exchange_unique_id(rank, world, make_id, tag='fokl'):
    """Rank 0 creates the 128-byte RCCL id; the others fetch it from a TCP store at MASTER_ADDR:MASTER_PORT."""
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500'))
    from torch.distributed import TCPStore
    import datetime
    # Under torch.distributed.run the elastic agent already serves a store on MASTER_PORT and tells its workers so;
    # then every rank (rank 0 included) connects as a client.
    agent_store = os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '') == 'True'
    store = TCPStore(addr, port, world, is_master=(rank == 0 and not agent_store),
                     timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
    key = tag + '_rccl_id_' + os.environ.get('TORCHELASTIC_RUN_ID', '0')
    if rank == 0:
        uid = make_id()
        store.set(key, uid)
    else:
        uid = store.get(key)
    return bytes(uid), store


class RcclComm:
    """RCCL communicator attached to a ``_capi.DeviceContext`` (backend "nccl" == RCCL on ROCm)."""

    def __init__(self, ctx, rank, world, unique_id=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._store = None
        if unique_id is None:
            unique_id, self._store = _exchange_unique_id(self.rank, self.world, ctx.comm_unique_id)
        ctx.comm_init(unique_id, self.rank, self.world)

    def allgather(self, values):
        return self.ctx.allgather(np.asarray(values, dtype=np.float64).reshape(-1), self.world)

    def allreduce_sum(self, values):
        return self.ctx.allreduce_sum(values)

    def barrier(self):
        self.ctx.allreduce_sum(np.zeros(1))
        self.ctx.sync()

    def close(self):
        self.ctx.comm_destroy()


class GlooComm:
    """Same interface over an initialised ``torch.distributed`` process group (CPU tests)."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def allgather(self, values):
        import torch
        v = torch.as_tensor(np.asarray(values, dtype=np.float64).reshape(-1))
        out = [torch.empty_like(v) for _ in range(self.world)]
        self._dist.all_gather(out, v)
        return np.stack([o.numpy() for o in out], axis=0)

    def allreduce_sum(self, values):
        import torch
        v = torch.as_tensor(np.array(values, dtype=np.float64, copy=True))
        self._dist.all_reduce(v)
        return v.numpy()

    def barrier(self):
        self._dist.barrier()

    def close(self):
        pass


def shard_range(count, rank, world):
    """Contiguous block partition of ``count`` units: the slice owned by ``rank``."""
    base, extra = divmod(int(count), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
