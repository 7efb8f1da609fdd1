#!/bin/bash
set -o pipefail
for mode in "b3s11 FOKL_X=1 FOKL_Y=1" "b5s9 FOKL_BULK_THREADS=5 FOKL_SPECTRAL_THREADS=9" "b6s8 FOKL_BULK_THREADS=6 FOKL_SPECTRAL_THREADS=8" "b4s10 FOKL_BULK_THREADS=4 FOKL_SPECTRAL_THREADS=10" "b5s11 FOKL_BULK_THREADS=5 FOKL_SPECTRAL_THREADS=11"; do
  set -- $mode
  QB_ARGS="--config 3 --steps 3 --warmup 1" bash tools/quick_bench.sh c3_$1 $2 $3 | cut -c1-40 || exit 1
  python - $1 <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_c3_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]; rs = d["random_stream"]
print("   ", sys.argv[1], "ms", round(d["ms_per_step"], 1), "walker wait", round(rs["walker_waiting_for_bulk_s_per_step"] * 1e3), "settle", round(h["t_settle"] * 1e3), "tests", round(h["phase_tests"] * 1e3), "cpu", round(d["cpu_seconds_per_step"], 2))
PY
done
