"""Worker for tests/test_dist_gloo.py: launched with torch.distributed.run (2 ranks, gloo, CPU).

Exercises the N > 1 host logic with the same communicator interface the GPU path uses (dist.RcclComm <-> GlooComm below):
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



# The product's communicators are RCCL over xGMI and a TCP control plane (fokl_gpy_amd/dist.py, no PyTorch); this stand-in
# with the same interface over a torch.distributed gloo group is test scaffolding and lives here.
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


def synth(seed, n, m):
    rng = np.random.default_rng(seed)
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.05 * rng.standard_normal(n)
    return x, y


def many_ranks(out_dir):
    """`dist_worker.py OUT many`: more ranks than a forward step has candidates.  M = 3 inputs, 2-way: the sub-stages bring
    T = 3 or 6 candidate terms (one per input or pair; two orders over a pair), so with 4 or 8 ranks T < world or T % world != 0 -- shares of one
    column on some ranks and PADDING-only shares on the others (engine._share_of).  Candidates and hybrid: model, calls, stream
    = the single-process fit's, ranks bitwise alike."""
    tdist.init_process_group('gloo')
    comm = GlooComm()
    rank, world = comm.rank, comm.world
    res = {'world': world}
    n, m = 640, 3
    rng = np.random.default_rng(23)
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.05 * rng.standard_normal(n)
    hy = dict(kernel='Bernoulli Polynomials', burnin=40, draws=40, a=4, atau=4, UserWarnings=False, ConsoleOutput=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        single = FoKLRoutines.FoKL(**hy)
        single._backend_override = OracleBackend()
        np.random.seed(4)
        sb, sm, se = single.fit(x, y, clean=True)
        single_state = np.random.get_state()
        res['new_terms'] = sorted({int(t['built']) for t in single.fit_trace if not t['kill']})
        for mode in ('cand', 'hybrid'):
            if mode == 'cand':
                fit = FoKLRoutines.FoKL(**hy)
                backend = OracleBackend()
                fit._backend_override = backend
                fit._prepare_fit(x, y, dict(clean=True))
                np.random.seed(4)
                fb, fm, fe = fit._search(backend, n, m, comm=comm, candidate_sharded=True)
            else:
                lo, hi = dist.shard_range(n, rank, world)
                fit = FoKLRoutines.FoKL(b=single.b, btau=single.btau, **hy)
                backend = ShardedOracleBackend(comm)
                fit.inputs, fit.data = single.inputs[lo:hi], single.data[lo:hi]
                fit._upload(backend, fit.inputs, fit.data)
                np.random.seed(4)
                fb, fm, fe = fit._search(backend, hi - lo, m, n_global=n, row_sharded=True, comm=comm, candidate_sharded=True)
            state = np.random.get_state()
            res[mode + '_driver'] = fit.fit_stats['search_driver']
            res[mode + '_mtx_equal'] = bool(fm.shape == sm.shape and np.array_equal(fm, sm))
            res[mode + '_calls_equal'] = [t['cols'] for t in fit.fit_trace] == [t['cols'] for t in single.fit_trace]
            res[mode + '_stream_equal'] = bool(np.array_equal(single_state[1], state[1]) and single_state[2:] == state[2:])
            res[mode + '_evs_err'] = float(np.max(np.abs(fe - se) / np.abs(se))) if len(fe) == len(se) else 1.0
            res[mode + '_betas_err'] = float(np.max(np.abs(fb - sb) / np.max(np.abs(sb), axis=0))) if fb.shape == sb.shape else 1.0
            res[mode + '_gathers'] = int(fit.fit_stats.get('candidate_gathers', 0))
            digest = comm.allgather([float(np.sum(fb)), float(np.sum(fe)), float(fm.sum())])
            res[mode + '_ranks_bitwise_equal'] = bool(np.all(digest == digest[0]))
    comm.barrier()
    with open(os.path.join(out_dir, f'rank{rank}.json'), 'w') as fh:
        json.dump(res, fh)
    tdist.destroy_process_group()


def main():
    out_dir = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == 'many':
        return many_ranks(out_dir)
    tdist.init_process_group('gloo')
    comm = GlooComm()
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
    res['rowshard_driver'] = sharded.fit_stats['search_driver']
    res['rowshard_calls_equal'] = [t['cols'] for t in sharded.fit_trace] == [t['cols'] for t in single.fit_trace]
    digest = comm.allgather([float(np.sum(rb)), float(np.sum(re)), float(rm.sum())])
    res['rowshard_ranks_bitwise_equal'] = bool(np.all(digest == digest[0]))

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

    # 4. candidate-sharded fit (north_star's split): every rank holds all rows and drives the same search from the same
    #    stream; G2 + BIC of the candidate models are dealt over the ranks and all-gathered.  3-way interactions so that
    #    sub-stages have many kill tests; must reproduce the single-process fit.
    n, m = 900, 5
    rng = np.random.default_rng(17)
    x = rng.random((n, m))
    y = np.sin(4 * x[:, 0]) + x[:, 1] * x[:, 2] * x[:, 3] + 0.4 * x[:, 4] ** 2 + 0.05 * rng.standard_normal(n)
    hy3 = dict(hy, way3=True, phis=getKernels.bernoulli()[:3])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        single = FoKLRoutines.FoKL(**hy3)
        single._backend_override = OracleBackend()
        np.random.seed(9)
        sb, sm, se = single.fit(x, y, clean=True)
        single_state = np.random.get_state()
        shard = FoKLRoutines.FoKL(**hy3)
        backend = OracleBackend()
        shard._backend_override = backend
        shard._prepare_fit(x, y, dict(clean=True))
        np.random.seed(9)
        cb, cm, ce = shard._search(backend, n, m, comm=comm, candidate_sharded=True)
        shard_state = np.random.get_state()
    res['cand_mtx_equal'] = bool(cm.shape == sm.shape and np.array_equal(cm, sm))
    res['cand_evs_err'] = float(np.max(np.abs(ce - se) / np.abs(se))) if len(ce) == len(se) else 1.0
    res['cand_betas_err'] = float(np.max(np.abs(cb - sb) / np.max(np.abs(sb), axis=0))) if cb.shape == sb.shape else 1.0
    res['cand_calls_equal'] = [t['cols'] for t in shard.fit_trace] == [t['cols'] for t in single.fit_trace]
    res['cand_stream_equal'] = bool(np.array_equal(single_state[1], shard_state[1]) and single_state[2:] == shard_state[2:])
    res['cand_driver'] = shard.fit_stats['search_driver']
    res['cand_gathers'] = int(shard.fit_stats.get('candidate_gathers', 0))
    res['cand_substages'] = int(shard.fit_stats['substages'])
    res['cand_gibbs_calls'] = int(shard.fit_stats['gibbs_calls'])
    # bitwise agreement between the ranks: what every rank ends up with
    digest = comm.allgather([float(np.sum(cb)), float(np.sum(ce)), float(cm.sum())])
    res['cand_ranks_bitwise_equal'] = bool(np.all(digest == digest[0]))

    # 4b. the same split on the Python statement of the loop (FOKL_SEARCH_DIST=python): G2 + BIC of the candidate models are
    #     dealt over the ranks and all-gathered window by window
    os.environ['FOKL_SEARCH_DIST'] = 'python'
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        shard = FoKLRoutines.FoKL(**hy3)
        backend = OracleBackend()
        shard._backend_override = backend
        shard._prepare_fit(x, y, dict(clean=True))
        np.random.seed(9)
        pb, pm, pe = shard._search(backend, n, m, comm=comm, candidate_sharded=True)
        py_state = np.random.get_state()
    del os.environ['FOKL_SEARCH_DIST']
    res['pycand_driver'] = shard.fit_stats['search_driver']
    res['pycand_mtx_equal'] = bool(pm.shape == sm.shape and np.array_equal(pm, sm))
    res['pycand_evs_err'] = float(np.max(np.abs(pe - se) / np.abs(se))) if len(pe) == len(se) else 1.0
    res['pycand_betas_err'] = float(np.max(np.abs(pb - sb) / np.max(np.abs(sb), axis=0))) if pb.shape == sb.shape else 1.0
    res['pycand_stream_equal'] = bool(np.array_equal(single_state[1], py_state[1]) and single_state[2:] == py_state[2:])
    res['pycand_remote'] = int(shard.fit_stats['spectral_remote'])
    res['pycand_exchanges'] = int(shard.fit_stats['exchanges'])
    res['pycand_gibbs_calls'] = int(shard.fit_stats['gibbs_calls'])
    digest = comm.allgather([float(np.sum(pb)), float(np.sum(pe)), float(pm.sum())])
    res['pycand_ranks_bitwise_equal'] = bool(np.all(digest == digest[0]))

    # 5. hybrid: rows AND candidates sharded (each rank holds half of the rows of section 4's dataset: its device work is
    #    halved; the eigen-decompositions of the replicated search are dealt over the ranks on top)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        lo, hi = dist.shard_range(n, rank, world)
        hyb = FoKLRoutines.FoKL(b=single.b, btau=single.btau, **hy3)
        backend = ShardedOracleBackend(comm)
        hyb.inputs, hyb.data = single.inputs[lo:hi], single.data[lo:hi]
        hyb._upload(backend, hyb.inputs, hyb.data)
        np.random.seed(9)
        hb, hm, he = hyb._search(backend, hi - lo, m, n_global=n, row_sharded=True, comm=comm, candidate_sharded=True)
        hyb_state = np.random.get_state()
    res['hybrid_mtx_equal'] = bool(hm.shape == sm.shape and np.array_equal(hm, sm))
    res['hybrid_evs_err'] = float(np.max(np.abs(he - se) / np.abs(se))) if len(he) == len(se) else 1.0
    res['hybrid_betas_err'] = float(np.max(np.abs(hb - sb) / np.max(np.abs(sb), axis=0))) if hb.shape == sb.shape else 1.0
    res['hybrid_calls_equal'] = [t['cols'] for t in hyb.fit_trace] == [t['cols'] for t in single.fit_trace]
    res['hybrid_stream_equal'] = bool(np.array_equal(single_state[1], hyb_state[1]) and single_state[2:] == hyb_state[2:])
    res['hybrid_driver'] = hyb.fit_stats['search_driver']
    digest = comm.allgather([float(np.sum(hb)), float(np.sum(he)), float(hm.sum())])
    res['hybrid_ranks_bitwise_equal'] = bool(np.all(digest == digest[0]))

    comm.barrier()
    with open(os.path.join(out_dir, f'rank{rank}.json'), 'w') as fh:
        json.dump(res, fh)
    tdist.destroy_process_group()


if __name__ == '__main__':
    main()
