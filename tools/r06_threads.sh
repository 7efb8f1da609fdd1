#!/bin/bash
# round 6, final tree: finish / chain / spectral thread counts of the headline fit once more (same box, two rounds)
set -o pipefail
for round in 1 2; do
for mode in "f2c2s8 FOKL_X=1 FOKL_Y=1 FOKL_Z=1" "f3c2s8 FOKL_FINISH_THREADS=3 FOKL_Y=1 FOKL_Z=1" "f4c2s8 FOKL_FINISH_THREADS=4 FOKL_Y=1 FOKL_Z=1" "f2c3s8 FOKL_CHAIN_THREADS=3 FOKL_Y=1 FOKL_Z=1" "f2c2s10 FOKL_SPECTRAL_THREADS=10 FOKL_Y=1 FOKL_Z=1" "f2c2s6 FOKL_SPECTRAL_THREADS=6 FOKL_Y=1 FOKL_Z=1" "f1c2s8 FOKL_FINISH_THREADS=1 FOKL_Y=1 FOKL_Z=1"; do
  set -- $mode
  bash tools/quick_bench.sh th_$1_$round $2 $3 $4 | cut -c1-40 || exit 1
  python - $1 $round <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_th_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]
print("     ", {k: round(h[k] * 1e3, 2) for k in ("t_eigh", "t_chain", "phase_model", "phase_statistics", "phase_tests", "t_final_verify")})
PY
done
done
