#!/bin/bash
# Round-6 profile collection on the GPU box (run through gpurun from the repo root).
# 1./2. PMC passes (FETCH_SIZE, WRITE_SIZE separately: the TCC block cannot hold both), kernel trace only -- never combined
# with sys/hip/hsa traces -> profiles/pmc_r06.json (bench.py reads it for `traffic`); 3. kernel trace + stats of the default
# bench command; 4. the driver's command un-profiled; 5. configs 1 / 3 / 4; 6. the host-side stream bench and the sweep of
# thread plans.
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
FLAGS="--steps 1 --warmup 0 --no-cpu-baseline --no-microbench --no-throughput"
FOKL_GRAM_TRACE=$OUT/gram_trace_fetch.txt rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $FLAGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
FOKL_GRAM_TRACE=$OUT/gram_trace_write.txt rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $FLAGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
python3 tools/rocprof_summary.py json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_r06.json 1000000 8 $OUT/gram_trace_fetch.txt $OUT/gram_trace_write.txt || exit 1
cp $OUT/pmc_r06.json profiles/pmc_r06.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-microbench --no-throughput > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.json 2> $OUT/bench_plain.err || exit 1
rm -f $OUT/summary.md
echo "## kernel trace + stats: python3 bench.py --steps 5 --warmup 2 --no-microbench --no-throughput" >> $OUT/summary.md
python3 tools/rocprof_summary.py stats $OUT/stats $OUT/summary.md
echo "## PMC FETCH_SIZE (KiB per dispatch, raw): python3 bench.py $FLAGS" >> $OUT/summary.md
python3 tools/rocprof_summary.py pmc $OUT/pmc_fetch $OUT/summary.md FETCH_SIZE
echo "## PMC WRITE_SIZE (KiB per dispatch): python3 bench.py $FLAGS" >> $OUT/summary.md
python3 tools/rocprof_summary.py pmc $OUT/pmc_write $OUT/summary.md WRITE_SIZE
cp $(ls $OUT/stats/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats.csv
for c in 1 3 4; do python3 bench.py --config $c --no-cpu-baseline --no-throughput --no-microbench > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
bash tools/stream_bench.sh $OUT/stream_bench.txt
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
