#!/bin/bash
# tools/stream_bench.sh [out]: builds tools/stream_bench.cpp and runs it over model sizes / thread counts on this host.
set -e
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/stream_bench.txt}
mkdir -p "$(dirname "$out")"
g++ -O3 -std=c++17 -ffp-contract=off -fno-math-errno -pthread -w -o /tmp/stream_bench tools/stream_bench.cpp
{
  lscpu | grep -E "Model name|^CPU\(s\)|Thread|L2|L3|MHz" || true
  nproc
  cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
  for p1 in 8 70 300; do
    for nt in 1 2 3 4; do /tmp/stream_bench $p1 $nt 100; done
  done
  FOKL_STREAM_SCALAR_WALK=1 /tmp/stream_bench 70 3 100
} > "$out" 2>&1
