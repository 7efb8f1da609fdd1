#!/bin/bash
# round 6: the producers' distance in front of the walker growing with the walker's waiting time, against the fixed 24 segments
set -o pipefail
pick() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]; rs = d["random_stream"]
print("     ms", round(d["ms_per_step"], 2), "bulk cpu ms", round(rs["bulk_threads_cpu_s_per_step"] * 1e3, 1), "segments", rs["segments_of_79872_doubles_per_step"], "walker busy", round(rs["walker_busy_s_per_step"] * 1e3, 1), "waiting for bulk", round(rs["walker_waiting_for_bulk_s_per_step"] * 1e3, 2), "settle", round(h.get("t_settle", 0) * 1e3, 1), "cpu", round(d["cpu_seconds_per_step"], 3))
PY
}
for round in 1 2; do
  for mode in "fixed FOKL_STREAM_AHEAD=fixed" "grows FOKL_STREAM_AHEAD=grows"; do
    set -- $mode
    bash tools/quick_bench.sh ah_$1_$round $2 | cut -c1-40 || exit 1
    pick ah_$1_$round
  done
done
for mode in "fixed FOKL_STREAM_AHEAD=fixed FOKL_Y=1" "grows FOKL_STREAM_AHEAD=grows FOKL_Y=1" "grows_b4 FOKL_BULK_THREADS=4 FOKL_Y=1" "grows_b5 FOKL_BULK_THREADS=5 FOKL_Y=1" "grows_b6 FOKL_BULK_THREADS=6 FOKL_Y=1" "grows_same FOKL_BULK_CPUS=same FOKL_Y=1" "fixed2 FOKL_STREAM_AHEAD=fixed FOKL_Y=1" "grows2 FOKL_X=1 FOKL_Y=1"; do
  set -- $mode
  QB_ARGS="--config 3 --steps 3 --warmup 1" bash tools/quick_bench.sh ah3_$1 $2 $3 | cut -c1-40 || exit 1
  pick ah3_$1
done
