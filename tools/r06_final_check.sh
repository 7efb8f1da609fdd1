#!/bin/bash
# round 6: the GPU suite, then the driver's bench command, on one box
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_r06b.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_r06b.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06b.json 2> gpurun_out/bench_r06b.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r06b.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("throughput_mode", {}).get("value"), d.get("parity", {}).get("ok"))
PY
