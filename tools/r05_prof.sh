#!/bin/bash
set -o pipefail
out=gpurun_out/r05c
mkdir -p $out
FOKL_SEARCH_PROFILE=1 timeout -k 10 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-microbench --no-throughput --no-parity > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
grep "fokl_search profile" $out/bench.err | tail -5
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05c/bench.json').read().strip().splitlines()[-1])
h = d.get('host_main_thread_s_per_step', {})
print('ms', round(d['ms_per_step'], 2), {k: round(v * 1e3, 2) if isinstance(v, float) and v < 10 else v for k, v in h.items()})
PY
