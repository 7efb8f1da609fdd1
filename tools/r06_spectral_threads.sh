#!/bin/bash
# round 6, final tree: spectral threads of the headline fit, interleaved, same box
set -o pipefail
for round in 1 2 3 4; do
for mode in "s8 FOKL_SPECTRAL_THREADS=8" "s6 FOKL_SPECTRAL_THREADS=6" "s5 FOKL_SPECTRAL_THREADS=5" "s4 FOKL_SPECTRAL_THREADS=4" "s7 FOKL_SPECTRAL_THREADS=7"; do
  set -- $mode
  bash tools/quick_bench.sh sp_$1_$round $2 | cut -c1-40 || exit 1
  python - $1 $round <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_sp_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]
print("     ", {k: round(h[k] * 1e3, 2) for k in ("t_eigh", "t_chain", "t_settle", "phase_tests", "t_final_verify", "t_search_body", "pool_spectral_s")})
PY
done
done
