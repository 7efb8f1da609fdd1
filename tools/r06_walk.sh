#!/bin/bash
# round 6: the headline fit with the walk in rank space against round 5's position walk, and the bulk thread count
set -o pipefail
mkdir -p gpurun_out
for mode in "ranked FOKL_X=1" "positions FOKL_STREAM_WALK=positions" "ranked6 FOKL_BULK_THREADS=6" "ranked8 FOKL_BULK_THREADS=8" "ranked_b FOKL_X=2" "positions_b FOKL_STREAM_WALK=positions"; do
  set -- $mode
  bash tools/quick_bench.sh r06w_$1 $2 || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/qb_r06w_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    rs = d.get('random_stream', {})
    print(f.split('qb_')[1], round(d['ms_per_step'], 2), 'walker', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rs.items()}, 'cpu', d['cpu_seconds_per_step_by_thread'])
PY
