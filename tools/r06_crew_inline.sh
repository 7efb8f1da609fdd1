#!/bin/bash
# round 6: the crew's walker walking the iteration at a segment's end from the presumed state instead of waiting for every verdict
set -o pipefail
for round in 1 2 3; do
for mode in "drain FOKL_CREW_INLINE=0" "inline FOKL_CREW_INLINE=1"; do
  set -- $mode
  FOKL_WALK_PROFILE=1 bash tools/quick_bench.sh ci_$1_$round $2 | cut -c1-36 || exit 1
  grep "rank walk" gpurun_out/qb_ci_$1_$round.err | tail -1 | cut -c13-170
  python - $1 $round <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_ci_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("     walker busy ms", round(d["random_stream"]["walker_busy_s_per_step"] * 1e3, 1), "cpu walker", round(d["cpu_seconds_per_step_by_thread"]["walker"] * 1e3, 1))
PY
done
done
