#!/bin/bash
out=gpurun_out/r05_head2; rm -rf $out; mkdir -p $out
for h in 1 0 1 0; do
  rm -f $out/trace.txt
  FOKL_HEAD_START=$h FOKL_POOL_TRACE=$out/trace.txt timeout -k 10 300 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-microbench --no-throughput --no-parity > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
  echo "== head start $h"
  for f in -2 -3 -4 -5; do python3 tools/fit_timeline.py $out/trace.txt --fit $f | grep -E "^fit|pool_up|substage 1 |full_evaluated 9|full_statistics" | head -5 | tr '\n' ' '; echo; done
done
