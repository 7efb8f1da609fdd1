#!/bin/bash
set -o pipefail
out=gpurun_out/r05i
mkdir -p $out
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench > $out/bench_$i.json 2> $out/bench_$i.err || { tail -20 $out/bench_$i.err; exit 1; }
python - $i <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05i/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
tm = d['throughput_mode']
print('ms', round(d['ms_per_step'], 2), 'cpu/fit', round(d['cpu_seconds_per_step'], 4), 'tp', round(tm['value']), 'ms/fit/proc', round(tm['ms_per_fit_per_process'], 1), tm['host_cpu'], 'worker cpu', round(tm['worker_s_per_fit']['cpu_s'], 4), 'worker s', round(tm['worker_s_per_fit']['seconds'], 4))
PY
done
