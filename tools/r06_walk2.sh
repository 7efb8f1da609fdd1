#!/bin/bash
# round 6: where the walker's time goes inside a fit (FOKL_WALK_PROFILE), and the host's cache topology
set -o pipefail
mkdir -p gpurun_out
lscpu | grep -E "Model name|Socket|Core|Thread|L2|L3|NUMA" > gpurun_out/r06_lscpu.txt
cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list >> gpurun_out/r06_lscpu.txt
FOKL_WALK_PROFILE=1 bash tools/quick_bench.sh r06p FOKL_X=1 || exit 1
grep "rank walk" gpurun_out/qb_r06p.err | tail -3
cat gpurun_out/r06_lscpu.txt
