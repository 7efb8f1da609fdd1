#!/bin/bash
# round 6: the tempered words of a segment through non-temporal stores (scratch tempered in place): the bulk phase alone, then
# inside the headline fit, same box, against ordinary stores; last the four-process throughput mode both ways
set -o pipefail
g++ -O3 -std=c++17 -ffp-contract=off -fno-math-errno -pthread -w -o /tmp/bulk_bench tools/bulk_bench.cpp || exit 1
for m in "FOKL_SEGMENT_STORES=cached" "FOKL_TEMPER_CHUNK=16" "FOKL_TEMPER_CHUNK=256" "FOKL_TEMPER_CHUNK=1024" "FOKL_TEMPER_CHUNK=4096" "FOKL_TEMPER_CHUNK=16384"; do
  echo "alone, $m: $(env $m /tmp/bulk_bench 400 | tail -1)"
done
pick() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("     bulk cpu ms", round(d["random_stream"]["bulk_threads_cpu_s_per_step"] * 1e3, 1), "walker busy", round(d["random_stream"]["walker_busy_s_per_step"] * 1e3, 1), "waiting for bulk", round(d["random_stream"]["walker_waiting_for_bulk_s_per_step"] * 1e3, 2))
PY
}
for round in 1 2 3; do
  for mode in "cached FOKL_SEGMENT_STORES=cached" "c256 FOKL_TEMPER_CHUNK=256" "c1024 FOKL_TEMPER_CHUNK=1024" "c4096 FOKL_TEMPER_CHUNK=4096"; do
    set -- $mode
    bash tools/quick_bench.sh ss_$1_$round $2 | cut -c1-50 || exit 1
    pick ss_$1_$round
  done
done
for mode in "cached FOKL_SEGMENT_STORES=cached" "c1024 FOKL_TEMPER_CHUNK=1024" "cached2 FOKL_SEGMENT_STORES=cached" "c1024b FOKL_TEMPER_CHUNK=1024"; do
  set -- $mode
  env $2 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-microbench > gpurun_out/ss_tp_$1.json 2> gpurun_out/ss_tp_$1.err || exit 1
  python - $1 <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ss_tp_{sys.argv[1]}.json").read().strip().splitlines()[-1])
t = d["throughput_mode"]
print("throughput", sys.argv[1], round(t["value"]), "ms per fit per process", round(t["ms_per_fit_per_process"], 1), "worker cpu_s", round(t["worker_s_per_fit"]["cpu_s"], 4), t["host_cpu"])
PY
done
