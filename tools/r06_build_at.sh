#!/bin/bash
# round 6: K1 + K2 of the coming sub-stage launched at the sub-stage's start (in front of the wait for the model's eigenpairs)
# against behind the model's evaluation; same box
set -o pipefail
for round in 1 2 3; do
for mode in "model FOKL_BUILD_AHEAD_AT=model" "start FOKL_BUILD_AHEAD_AT=start"; do
  set -- $mode
  bash tools/quick_bench.sh ba_$1_$round $2 | cut -c1-40 || exit 1
  python - $1 $round <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_ba_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]
print("     ", {k: round(h[k] * 1e3, 2) for k in ("t_eigh", "t_resid", "t_chain", "phase_model", "phase_statistics", "phase_tests", "t_search_body")}, "early", h.get("forecasts_early"), "used", h.get("forecasts_used"))
PY
done
done
