#!/bin/bash
# round 6: where the serial section of the stream's production goes (configs[3] and the headline fit)
FOKL_WALK_PROFILE=1 QB_ARGS="--config 3 --steps 2 --warmup 1" bash tools/quick_bench.sh tk3 FOKL_X=1 | cut -c1-40
grep "segments; per segment" gpurun_out/qb_tk3.err | tail -3
FOKL_WALK_PROFILE=1 QB_ARGS="--config 3 --steps 2 --warmup 1" bash tools/quick_bench.sh tk3b FOKL_BULK_THREADS=6 | cut -c1-40
grep "segments; per segment" gpurun_out/qb_tk3b.err | tail -3
FOKL_WALK_PROFILE=1 bash tools/quick_bench.sh tk2 FOKL_X=1 | cut -c1-40
grep "segments; per segment" gpurun_out/qb_tk2.err | tail -3
