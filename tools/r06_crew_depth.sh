#!/bin/bash
set -o pipefail
for round in 1 2; do
for mode in "d8h2 FOKL_CREW_DEPTH=8 FOKL_WALK_HELPERS=2" "d16h2 FOKL_CREW_DEPTH=16 FOKL_WALK_HELPERS=2" "d4h2 FOKL_CREW_DEPTH=4 FOKL_WALK_HELPERS=2" "d16h3 FOKL_CREW_DEPTH=16 FOKL_WALK_HELPERS=3" "d12h3 FOKL_CREW_DEPTH=12 FOKL_WALK_HELPERS=3" "d16h4 FOKL_CREW_DEPTH=16 FOKL_WALK_HELPERS=4"; do
  set -- $mode
  FOKL_WALK_PROFILE=1 bash tools/quick_bench.sh cd_$1_$round $2 $3 | cut -c1-36 || exit 1
  grep "walk crew" gpurun_out/qb_cd_$1_$round.err | tail -1 | cut -c13-170
  python - $1 $round <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_cd_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("     walker busy ms", round(d["random_stream"]["walker_busy_s_per_step"] * 1e3, 1), "cpu walker", round(d["cpu_seconds_per_step_by_thread"]["walker"] * 1e3, 1))
PY
done
done
