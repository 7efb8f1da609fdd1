#!/bin/bash
set -o pipefail
out=gpurun_out/r05d
mkdir -p $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench $EXTRA > $out/bench_$name.json 2> $out/bench_$name.err || { tail -20 $out/bench_$name.err; exit 1; }
  python - $name <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05d/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
tm = d.get('throughput_mode') or {}
print(sys.argv[1], 'ms', round(d['ms_per_step'], 2), 'parity', (d.get('parity') or {}).get('ok'), 'cpu/fit', round(d['cpu_seconds_per_step'], 4),
      {k: round(v * 1e3, 1) for k, v in d['cpu_seconds_per_step_by_thread'].items()},
      'tp', round(tm.get('value', 0)), 'ms/fit/proc', round(tm.get('ms_per_fit_per_process', 0), 1), tm.get('host_cpu'), 'worker cpu', (tm.get('worker_s_per_fit') or {}).get('cpu_s'))
PY
}
run default FOKL_X=1
run depth12 FOKL_EIGH_UPDATE_DEPTH=12
run depth24 FOKL_EIGH_UPDATE_DEPTH=24
