#!/bin/bash
# round 6: the bulk threads on physical cores of their own (outside the driver's cache domain) against where they were; same box
set -o pipefail
pick() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]; rs = d["random_stream"]
print("     ms", round(d["ms_per_step"], 2), "bulk cpu ms", round(rs["bulk_threads_cpu_s_per_step"] * 1e3, 1), "walker busy", round(rs["walker_busy_s_per_step"] * 1e3, 1), "waiting for bulk", round(rs["walker_waiting_for_bulk_s_per_step"] * 1e3, 2), "settle", round(h.get("t_settle", 0) * 1e3, 1), "cpu", round(d["cpu_seconds_per_step"], 3))
PY
}
for round in 1 2 3; do
  for mode in "same FOKL_BULK_CPUS=same" "other FOKL_BULK_CPUS=other"; do
    set -- $mode
    bash tools/quick_bench.sh bp_$1_$round $2 | cut -c1-40 || exit 1
    pick bp_$1_$round
  done
done
for mode in "same FOKL_BULK_CPUS=same FOKL_X=1 FOKL_Y=1" "other FOKL_BULK_CPUS=other FOKL_X=1 FOKL_Y=1" "other_b5 FOKL_BULK_CPUS=other FOKL_BULK_THREADS=5 FOKL_Y=1" "other_b6 FOKL_BULK_CPUS=other FOKL_BULK_THREADS=6 FOKL_Y=1" "other_b8 FOKL_BULK_CPUS=other FOKL_BULK_THREADS=8 FOKL_Y=1" "other_b6_h0 FOKL_BULK_CPUS=other FOKL_BULK_THREADS=6 FOKL_WALK_HELPERS=0" "other_b6_s9 FOKL_BULK_CPUS=other FOKL_BULK_THREADS=6 FOKL_SPECTRAL_THREADS=9" "same2 FOKL_BULK_CPUS=same FOKL_X=1 FOKL_Y=1"; do
  set -- $mode
  QB_ARGS="--config 3 --steps 3 --warmup 1" bash tools/quick_bench.sh bp3_$1 $2 $3 $4 | cut -c1-40 || exit 1
  pick bp3_$1
done
