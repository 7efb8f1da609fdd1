#!/bin/bash
set -o pipefail
out=gpurun_out/r05k; mkdir -p $out
for spec in "w2=FOKL_X=1" "w6=FOKL_EIGH_UPDATE_DEPTH_WIDE=6" "w12=FOKL_EIGH_UPDATE_DEPTH_WIDE=12" "nodefer=FOKL_G2_DEFER_FROM=100000" "w6d24=FOKL_EIGH_UPDATE_DEPTH_WIDE=24"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs timeout -k 10 300 python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-microbench --no-throughput > $out/cfg3_$name.json 2> $out/cfg3_$name.err || { tail -5 $out/cfg3_$name.err; exit 1; }
  python - $name <<'PY'
import json, sys
d = json.loads([l for l in open(f'gpurun_out/r05k/cfg3_{sys.argv[1]}.json').read().strip().splitlines() if l.startswith('{')][-1])
h = d['host_main_thread_s_per_step']
print(sys.argv[1], 'ms', round(d['ms_per_step'], 1), 'parity', d['parity']['ok'], d['parity']['max_draw_err_over_scale'], 'settle', round(h['t_settle'] * 1e3), 'spectral cpu', round(h['pool_spectral_s'], 2), 'submitted', h['spectral_submitted'], 'updated', h['spectral_updated'], 'cpu', round(d['cpu_seconds_per_step'], 2))
PY
done
