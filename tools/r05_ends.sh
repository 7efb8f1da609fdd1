#!/bin/bash
# where a fit's first and last milliseconds go: pool creation, the final confirmation of guessed decisions, teardown
out=gpurun_out/r05_ends; rm -rf $out; mkdir -p $out
timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-microbench --no-throughput > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
FOKL_SEARCH_PROFILE=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-microbench --no-throughput --no-parity > $out/bench_prof.json 2> $out/bench_prof.err || { tail -20 $out/bench_prof.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_ends/bench.json').read().strip().splitlines()[-1])
h = d.get('host_main_thread_s_per_step', {})
print('ms', round(d['ms_per_step'], 2), {k: round(v * 1e3, 2) for k, v in h.items() if k.startswith('t_') or k.startswith('phase')})
PY
grep -v "^$" $out/bench_prof.err | tail -60 | cut -c1-400
