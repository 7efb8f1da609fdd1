"""
Multi-GPU plumbing: one process per GPU, RCCL over xGMI through the C ABI (``fokl_comm_*``).

The forward-selection path has exactly two exchange steps, both tiny:
  * all-gather of per-candidate BIC values when kill-test proposals (or whole independent fits) are sharded
    over ranks;
  * all-reduce(sum) of Gram blocks / residual moments when *rows* are sharded over ranks.

``RcclComm`` drives them on the GPU; ``GlooComm`` offers the same interface over ``torch.distributed``'s gloo
backend so that the N > 1 host logic is testable on CPU (tests/test_dist_gloo.py).  torch is used for that CPU
test only: a GPU process never imports it (its bundled HIP / RCCL copies must not share a process with ours).
"""
import os

import numpy as np


def env_rank_world():
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    return rank, world, local


def flush_c_streams():
    """Flush C stdio buffers (librccl writes its banner there) so that they cannot land after later Python output."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


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


def _exchange_unique_id(rank, world, make_id, tag='fokl', timeout_s=300.0):
    """
    Rank 0 creates the 128-byte RCCL id, the other ranks of the node read it.

    One node, one process per GPU (the launch contract), so the hand-off is a file in the node's temp directory,
    written atomically and named after what all ranks of one launch share: MASTER_PORT and the launcher's pid.
    Deliberately NOT torch.distributed's TCPStore: importing torch loads its private copies of the HIP runtime and
    of librccl into the process, and RCCL then initialises against the wrong runtime ("unhandled cuda error").
    """
    import tempfile
    import time
    port = os.environ.get('MASTER_PORT', '0')
    launch = os.environ.get('TORCHELASTIC_RUN_ID', 'none')
    path = os.path.join(tempfile.gettempdir(), f'{tag}_rccl_id_{port}_{launch}_{os.getppid()}.bin')
    if rank == 0:
        uid = bytes(make_id())
        tmp = path + f'.{os.getpid()}.tmp'
        with open(tmp, 'wb') as fh:
            fh.write(uid)
        os.replace(tmp, path)
        return uid, path
    deadline = time.monotonic() + timeout_s
    while True:
        try:
            with open(path, 'rb') as fh:
                uid = fh.read()
            if len(uid) == 128:
                return uid, path
        except FileNotFoundError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no RCCL id from rank 0 at {path} after {timeout_s:.0f} s")
        time.sleep(0.01)


class RcclComm:
    """RCCL communicator attached to a ``_capi.DeviceContext`` (backend "nccl" == RCCL on ROCm)."""

    def __init__(self, ctx, rank, world, unique_id=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._id_file = None
        if unique_id is None:
            unique_id, self._id_file = _exchange_unique_id(self.rank, self.world, ctx.comm_unique_id)
        # librccl prints a version banner on C stdout during init; benchmark drivers parse stdout, so route file
        # descriptor 1 to stderr for the duration of the call
        import sys
        sys.stdout.flush()
        flush_c_streams()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            ctx.comm_init(unique_id, self.rank, self.world)  # collective: returns once every rank has joined
            flush_c_streams()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        if self._id_file is not None and self.rank == 0:
            try:
                os.remove(self._id_file)                      # everybody has read it by now
            except OSError:
                pass

    def allgather(self, values):
        return self.ctx.allgather(np.asarray(values, dtype=np.float64).reshape(-1), self.world)

    def allreduce_sum(self, values):
        return self.ctx.allreduce_sum(values)

    def barrier(self):
        self.ctx.allreduce_sum(np.zeros(1))
        self.ctx.sync()

    def close(self):
        self.ctx.comm_destroy()
        flush_c_streams()


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
