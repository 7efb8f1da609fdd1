#!/bin/bash
set -o pipefail
out=gpurun_out/r05g
mkdir -p $out
for c in 1 3 4; do
  timeout -k 10 500 python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-microbench --no-throughput > $out/bench_cfg$c.json 2> $out/bench_cfg$c.err || { tail -30 $out/bench_cfg$c.err; exit 1; }
  python - $c <<'PY'
import json, sys
d = json.loads([l for l in open(f'gpurun_out/r05g/bench_cfg{sys.argv[1]}.json').read().strip().splitlines() if l.startswith('{')][-1])
h = d.get('host_main_thread_s_per_step') or {}
print('cfg', sys.argv[1], 'ms', round(d['ms_per_step'], 1), 'value', round(d['value']), 'parity', (d.get('parity') or {}).get('ok'), (d.get('parity') or {}).get('max_draw_err_over_scale'), d.get('kill_decisions'), 'cpu', round(d.get('cpu_seconds_per_step', 0), 3), d['config'].get('workload', '')[:80])
PY
done
