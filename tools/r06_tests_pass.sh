#!/bin/bash
# round 6: what the walk's passes cost per iteration inside the headline fit: segment words cached / past the cache, bulk threads
# here / elsewhere, prefetch of the second gamma site
set -o pipefail
for round in 1 2; do
for mode in "nt_other FOKL_X=1 FOKL_Y=1 FOKL_Z=1" "nt_other_pf1 FOKL_WALK_PREFETCH=first FOKL_Y=1 FOKL_Z=1" "nt_same FOKL_BULK_CPUS=same FOKL_Y=1 FOKL_Z=1" "cached_same FOKL_BULK_CPUS=same FOKL_SEGMENT_STORES=cached FOKL_Z=1" "cached_other FOKL_SEGMENT_STORES=cached FOKL_Y=1 FOKL_Z=1" "nt_other_h3 FOKL_WALK_HELPERS=3 FOKL_Y=1 FOKL_Z=1"; do
  set -- $mode
  FOKL_WALK_PROFILE=1 bash tools/quick_bench.sh tp_$1_$round $2 $3 $4 | cut -c1-36 || exit 1
  grep "rank walk" gpurun_out/qb_tp_$1_$round.err | tail -1 | cut -c13-130
  python - $1 $round <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_tp_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("     walker busy ms", round(d["random_stream"]["walker_busy_s_per_step"] * 1e3, 1), "bulk", round(d["random_stream"]["bulk_threads_cpu_s_per_step"] * 1e3, 1))
PY
done
done
