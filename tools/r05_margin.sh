#!/bin/bash
set -o pipefail
out=gpurun_out/r05h
mkdir -p $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench --no-throughput > $out/bench_$name.json 2> $out/bench_$name.err || { tail -20 $out/bench_$name.err; exit 1; }
  python - $name <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05h/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
h = d['host_main_thread_s_per_step']
print(f"{sys.argv[1]:12s} ms {d['ms_per_step']:.2f} parity {d['parity']['ok']}", {k: round(h[k] * 1e3, 2) for k in ('phase_tests', 't_final_verify', 't_eigh', 't_settle', 't_chain', 'pool_spectral_s')}, 'guess_waits', h['guess_waits'], 'guessed', h['guessed'], d['kill_decisions']['max_rel_distance_of_guessed_intercept_scale'], 'cpu', round(d['cpu_seconds_per_step'], 4))
PY
}
run default FOKL_X=1
run margin005 FOKL_GUESS_MARGIN=0.005
run margin002 FOKL_GUESS_MARGIN=0.002
run default2 FOKL_X=1
