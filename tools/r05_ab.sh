#!/bin/bash
# usage: tools/r05_ab.sh NAME=ENV ... : bench of configs[2] per environment setting, interleaved twice
set -o pipefail
out=gpurun_out/r05j
mkdir -p $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench --no-throughput > $out/bench_$name.json 2> $out/bench_$name.err || { tail -20 $out/bench_$name.err; exit 1; }
  python - $name <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05j/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
h = d['host_main_thread_s_per_step']
print(f"{sys.argv[1]:14s} ms {d['ms_per_step']:.2f} parity {d['parity']['ok']}", {k: round(h[k] * 1e3, 2) for k in ('phase_prepare', 'phase_model', 'phase_statistics', 'phase_tests', 't_final_verify', 't_eigh', 'pool_noise_s', 'noise_verdict_wait_s', 'noise_queue_wait_s', 'pool_spectral_s')}, 'fc_used', h.get('forecasts_used'), 'kill_loop', round(h['t_kill_loop']*1e3,2), 'cpu', round(d['cpu_seconds_per_step'], 4))
PY
}
for pass in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    run ${name}_$pass $envs
  done
done
