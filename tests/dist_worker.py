"""Worker for tests/test_dist_gloo.py: launched with torch.distributed.run (2 ranks, gloo, CPU).

Exercises the N > 1 host logic with the same communicator interface the GPU path uses (dist.RcclComm <-> dist.GlooComm):
  1. the two collectives (all-gather, all-reduce-sum) and the shard partition;
  2. a ROW-SHARDED fit: each rank holds half of the rows, Gram blocks / residual moments are summed over ranks,
     the (N-independent) sampler is replicated -> every rank must select the same model as a single-process fit;
  3. the bench-style throughput mode: independent fits per rank, one all-gather of the per-rank counters.
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch.distributed as tdist  # noqa: E402

from fokl_gpy_amd import FoKLRoutines, dist, getKernels  # noqa: E402
from helpers import OracleBackend  # noqa: E402


class ShardedOracleBackend(OracleBackend):
    """OracleBackend on a row shard; partial sums are combined through the communicator like fokl_gram(allreduce=1)."""

    def __init__(self, comm):
        super().__init__()
        self.comm = comm

    def gram(self, row_slots, col_slots, allreduce=False):
        g = super().gram(row_slots, col_slots)
        return self.comm.allreduce_sum(g) if allreduce else g

    def bic_resid(self, slots, betahat, allreduce=False):
        s = np.array(super().bic_resid(slots, betahat))
        if allreduce:
            s = self.comm.allreduce_sum(s)
        return float(s[0]), float(s[1])


def synth(seed, n, m):
    rng = np.random.default_rng(seed)
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.05 * rng.standard_normal(n)
    return x, y


def main():
    out_dir = sys.argv[1]
    tdist.init_process_group('gloo')
    comm = dist.GlooComm()
    rank, world = comm.rank, comm.world
    res = {}

    # 1. collectives
    g = comm.allgather([rank + 0.5, 10.0 * rank])
    res['allgather_ok'] = bool(g.shape == (world, 2) and np.array_equal(g[:, 0], np.arange(world) + 0.5))
    r = comm.allreduce_sum(np.array([[1.0, rank], [2.0, 3.0]]))
    res['allreduce_ok'] = bool(np.array_equal(r, [[world, sum(range(world))], [2.0 * world, 3.0 * world]]))
    spans = [dist.shard_range(10, k, 3) for k in range(3)]
    res['shard_ok'] = spans == [(0, 4), (4, 7), (7, 10)]

    # 2. row-sharded fit vs single-process fit
    n, m = 1200, 3
    x, y = synth(5, n, m)
    hy = dict(kernel='Bernoulli Polynomials', burnin=80, draws=80, a=4, atau=4, UserWarnings=False, ConsoleOutput=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        single = FoKLRoutines.FoKL(**hy)
        single._backend_override = OracleBackend()
        np.random.seed(3)
        sb, sm, se = single.fit(x, y, clean=True)

        lo, hi = dist.shard_range(n, rank, world)
        sharded = FoKLRoutines.FoKL(b=single.b, btau=single.btau, **hy)
        backend = ShardedOracleBackend(comm)
        sharded.inputs, sharded.data = single.inputs[lo:hi], single.data[lo:hi]
        sharded._upload(backend, sharded.inputs, sharded.data)
        np.random.seed(3)
        rb, rm, re = sharded._search(backend, hi - lo, m, n_global=n, row_sharded=True)
    res['rowshard_mtx_equal'] = bool(rm.shape == sm.shape and np.array_equal(rm, sm))
    res['rowshard_evs_err'] = float(np.max(np.abs(re - se) / np.abs(se))) if len(re) == len(se) else 1.0
    res['rowshard_betas_err'] = float(np.max(np.abs(rb - sb) / np.max(np.abs(sb), axis=0))) if rb.shape == sb.shape else 1.0

    # 3. throughput mode: independent fits, one all-gather
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        xi, yi = synth(100 + rank, 600, 3)
        mine = FoKLRoutines.FoKL(**hy)
        mine._backend_override = OracleBackend()
        np.random.seed(1000 + rank)
        mine.fit(xi, yi, clean=True)
    gathered = comm.allgather([mine.fit_stats['terms_logical'], float(np.min(mine.evs))])
    res['replica_terms'] = gathered[:, 0].tolist()
    res['replica_best_bic'] = gathered[:, 1].tolist()
    res['replica_own_terms'] = mine.fit_stats['terms_logical']

    comm.barrier()
    with open(os.path.join(out_dir, f'rank{rank}.json'), 'w') as fh:
        json.dump(res, fh)
    tdist.destroy_process_group()


if __name__ == '__main__':
    main()
