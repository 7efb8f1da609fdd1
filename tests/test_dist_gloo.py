"""N > 1 host logic on CPU: two ranks, gloo backend, launched exactly like the driver launches bench.py; four and eight
ranks on a dataset whose forward steps have fewer candidates than there are ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

from helpers import ROOT


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_collectives_rowshard_and_replicas(tmp_path):
    env = dict(os.environ, OPENBLAS_NUM_THREADS='2', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py'), str(tmp_path)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    results = [json.load(open(tmp_path / f'rank{r}.json')) for r in range(2)]
    for res in results:
        assert res['allgather_ok'] and res['allreduce_ok'] and res['shard_ok']
        # row-sharded fit, native driver: same model and call sequence on every rank as the single-process fit (sums differ
        # only in association), the ranks agree to the last bit
        assert res['rowshard_driver'] == 'native'
        assert res['rowshard_mtx_equal'] and res['rowshard_calls_equal'] and res['rowshard_ranks_bitwise_equal']
        assert res['rowshard_evs_err'] < 1e-10 and res['rowshard_betas_err'] < 1e-8
        # candidate-sharded fit, native driver (round 5): every rank repeats the search, the Gram rows of a forward step's
        # candidate columns are computed one share per rank and all-gathered -- one gather per sub-stage; same model, call
        # sequence, stream and (to rounding) numbers as the single-process fit, ranks bitwise alike
        assert res['cand_driver'] == 'native'
        assert res['cand_mtx_equal'] and res['cand_calls_equal'] and res['cand_stream_equal']
        assert res['cand_evs_err'] < 1e-10 and res['cand_betas_err'] < 1e-8
        assert res['cand_ranks_bitwise_equal']
        assert res['cand_gathers'] >= res['cand_substages']
        # the Python loop's form of the split (FOKL_SEARCH_DIST=python): G2 + BIC of the candidate models dealt over the
        # ranks; about half of the spectral results arrived from the other rank
        assert res['pycand_driver'] == 'python'
        assert res['pycand_mtx_equal'] and res['pycand_stream_equal'] and res['pycand_ranks_bitwise_equal']
        assert res['pycand_evs_err'] < 1e-10 and res['pycand_betas_err'] < 1e-8
        assert res['pycand_exchanges'] > 0 and res['pycand_remote'] >= res['pycand_gibbs_calls'] // 2 - 2
        # hybrid (rows + candidates): under the native driver the rows are sharded (the Gram rows of a share of the
        # candidates over a share of the rows would be partial sums nobody can use); same model, calls and stream
        assert res['hybrid_driver'] == 'native'
        assert res['hybrid_mtx_equal'] and res['hybrid_calls_equal'] and res['hybrid_stream_equal']
        assert res['hybrid_evs_err'] < 1e-10 and res['hybrid_betas_err'] < 1e-8
        assert res['hybrid_ranks_bitwise_equal']
    # throughput mode: every rank sees all ranks' counters after the single all-gather
    assert results[0]['replica_terms'] == results[1]['replica_terms']
    assert results[0]['replica_terms'][0] == results[0]['replica_own_terms']
    assert results[1]['replica_terms'][1] == results[1]['replica_own_terms']


@pytest.mark.parametrize('world', [4, 8])
def test_more_ranks_than_candidates(tmp_path, world):
    """T = 3 or 6 candidate terms per forward step over 4 and 8 ranks (T < world, T % world != 0): ranks whose share of the
    candidates is padding only (engine._share_of); candidates and hybrid (rows + candidates) under the native driver --
    the single-process model, call sequence and stream, every rank the same bits."""
    env = dict(os.environ, OPENBLAS_NUM_THREADS='1', OMP_NUM_THREADS='1', FOKL_SPIN='0.05')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr',
           '127.0.0.1', '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py'), str(tmp_path), 'many']
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    results = [json.load(open(tmp_path / f'rank{r}.json')) for r in range(world)]
    for res in results:
        # forward steps of 3 and of 6 candidates: fewer than ranks, or not a multiple of their number
        assert res['world'] == world and res['new_terms'] == [3, 6]
        for mode in ('cand', 'hybrid'):
            assert res[mode + '_driver'] == 'native'
            assert res[mode + '_mtx_equal'] and res[mode + '_calls_equal'] and res[mode + '_stream_equal'], mode
            assert res[mode + '_evs_err'] < 1e-10 and res[mode + '_betas_err'] < 1e-8, mode
            assert res[mode + '_ranks_bitwise_equal'], mode
        assert res['cand_gathers'] >= 1
