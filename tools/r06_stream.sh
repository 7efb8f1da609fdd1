#!/bin/bash
# round 6: the walk in rank space against round 5's position walk on the GPU box's host (EPYC), per-pass cycle counts
set -e
out=gpurun_out/r06_stream.txt
mkdir -p gpurun_out
g++ -O3 -std=c++17 -ffp-contract=off -fno-math-errno -pthread -w -o /tmp/stream_bench tools/stream_bench.cpp
g++ -O3 -std=c++17 -DFOKL_WALK_PROFILE -ffp-contract=off -fno-math-errno -pthread -w -o /tmp/stream_bench_p tools/stream_bench.cpp
{
  lscpu | grep -E "Model name|MHz" || true
  for p1 in 8 30 70 110 150 300; do
    echo "== p1 $p1: ranked"; /tmp/stream_bench $p1 4 100 | tail -5
    echo "== p1 $p1: positions (round 5)"; FOKL_STREAM_WALK=positions /tmp/stream_bench $p1 4 100 | tail -5
    echo "== p1 $p1: ranked, per pass"; /tmp/stream_bench_p $p1 4 100 | tail -2
  done
} > $out 2>&1
cat $out
