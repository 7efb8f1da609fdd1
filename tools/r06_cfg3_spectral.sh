#!/bin/bash
# round 6: configs[3] after the stream's production got out of the way: spectral threads x bulk threads, same box
set -o pipefail
for mode in "s11b3 FOKL_SPECTRAL_THREADS=11 FOKL_BULK_THREADS=3" "s13b3 FOKL_SPECTRAL_THREADS=13 FOKL_BULK_THREADS=3" "s15b3 FOKL_SPECTRAL_THREADS=15 FOKL_BULK_THREADS=3" "s13b2 FOKL_SPECTRAL_THREADS=13 FOKL_BULK_THREADS=2" "s18b2 FOKL_SPECTRAL_THREADS=18 FOKL_BULK_THREADS=2" "s11b3again FOKL_SPECTRAL_THREADS=11 FOKL_BULK_THREADS=3"; do
  set -- $mode
  QB_ARGS="--config 3 --steps 3 --warmup 1" bash tools/quick_bench.sh c3s_$1 $2 $3 | cut -c1-40 || exit 1
  python - $1 <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_c3s_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]; rs = d["random_stream"]
print("   ", sys.argv[1], "ms", round(d["ms_per_step"], 1), "walker wait", round(rs["walker_waiting_for_bulk_s_per_step"] * 1e3), "settle", round(h["t_settle"] * 1e3), "t_eigh", round(h["t_eigh"] * 1e3), "tests", round(h["phase_tests"] * 1e3), "spectral busy", round(h["pool_spectral_s"], 2), "cpu", round(d["cpu_seconds_per_step"], 2))
PY
done
