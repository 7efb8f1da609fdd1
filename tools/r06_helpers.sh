#!/bin/bash
# round 6: helper threads of the walk inside the headline fit (native sub-stage loop): count x placement, same box
set -o pipefail
for round in 1 2; do
  for mode in "h0 FOKL_WALK_HELPERS=0 FOKL_X=1" "h3other FOKL_WALK_HELPERS=3 FOKL_WALK_CPUS=other" "h2other FOKL_WALK_HELPERS=2 FOKL_WALK_CPUS=other" "h1other FOKL_WALK_HELPERS=1 FOKL_WALK_CPUS=other" "h4other FOKL_WALK_HELPERS=4 FOKL_WALK_CPUS=other"; do
    set -- $mode
    FOKL_WALK_PROFILE=1 bash tools/quick_bench.sh hp_$1_$round $2 $3 | cut -c1-46 || exit 1
    grep "rank walk" gpurun_out/qb_hp_$1_$round.err | tail -1 | cut -c1-150
    python - $1 $round <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_hp_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("     walker busy ms", round(d["random_stream"]["walker_busy_s_per_step"] * 1e3, 1), "cpu", d["cpu_seconds_per_step_by_thread"]["walker"])
PY
  done
done
