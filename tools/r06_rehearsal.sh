#!/bin/bash
# round 6: `python bench.py --gpus 2` launching its own ranks (no torch launcher), both on the one GPU of a gpurun box
# (FOKL_BENCH_SHARE_GPU=1; RCCL refuses two ranks on a device: the exchange steps go over the TCP control plane,
# FOKL_BENCH_SHARDED_OVER_TCP=1): configs[3] solo, hybrid (rows + candidates) and candidates; configs[2] replicas + joint fit
set -o pipefail
out=gpurun_out/r06_rehearsal; rm -rf $out; mkdir -p $out
common="--no-cpu-baseline --no-microbench --no-throughput"
timeout -k 10 400 python bench.py --config 3 --steps 3 --warmup 1 $common > $out/cfg3_solo.json 2> $out/cfg3_solo.err || { tail -20 $out/cfg3_solo.err; exit 1; }
export FOKL_BENCH_SHARE_GPU=1 FOKL_BENCH_SHARDED_OVER_TCP=1
for mode in hybrid candidates; do
  timeout -k 10 500 python bench.py --gpus 2 --config 3 --mode $mode --steps 3 --warmup 1 $common > $out/cfg3_$mode.json 2> $out/cfg3_$mode.err || { tail -30 $out/cfg3_$mode.err; exit 1; }
done
timeout -k 10 400 python bench.py --gpus 2 --steps 8 --warmup 3 --no-cpu-baseline --no-microbench > $out/cfg2_2ranks.json 2> $out/cfg2_2ranks.err || { tail -30 $out/cfg2_2ranks.err; exit 1; }
python - <<'PY'
import json
def last(path):
    return json.loads([l for l in open(path).read().strip().splitlines() if l.startswith('{')][-1])
for name in ('cfg3_solo', 'cfg3_hybrid', 'cfg3_candidates', 'cfg2_2ranks'):
    d = last(f'gpurun_out/r06_rehearsal/{name}.json')
    print(name, 'ms_per_step', round(d['ms_per_step'], 1), 'n_gpus', d['n_gpus'], 'mode', d['config'].get('parallelism'),
          'driver', d.get('search_driver'), 'parity', (d.get('parity') or {}).get('ok'), 'collectives', d['config'].get('collectives'),
          'cpu', d.get('cpu_seconds_per_step'))
    if 'candidate_sharded' in d:
        print('   joint:', json.dumps(d['candidate_sharded'])[:400])
PY
