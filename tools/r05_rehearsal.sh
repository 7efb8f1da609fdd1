#!/bin/bash
# round 5: the driver's N-GPU command with two ranks on the one GPU of a gpurun box (FOKL_BENCH_SHARE_GPU=1; RCCL refuses two
# ranks on one device, the TCP control plane carries the collectives): replicas + the candidate-sharded joint fit, configs[2]
set -o pipefail
out=gpurun_out/r05f
mkdir -p $out
export FOKL_BENCH_SHARE_GPU=1 FOKL_BENCH_SHARDED_OVER_TCP=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 8 --warmup 3 --no-cpu-baseline --no-microbench > $out/bench_2ranks.json 2> $out/bench_2ranks.err || { tail -30 $out/bench_2ranks.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05f/bench_2ranks.json').read().strip().splitlines() if l.startswith('{')][-1])
print('replicas: value', round(d['value']), 'ms_per_step', round(d['ms_per_step'], 2), 'n_gpus', d['n_gpus'])
print('joint:', json.dumps(d.get('candidate_sharded')))
PY
