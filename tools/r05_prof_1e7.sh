#!/bin/bash
# per-kernel table of the configs[2]-family fit where the device binds (N = 1e7): kernel trace + stats only
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r05_1e7; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --rows 10000000 --steps 3 --warmup 2 --no-cpu-baseline --no-microbench --no-throughput --no-parity > $OUT/bench.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
cd $GRAFT_REPO_ROOT
echo "## kernel trace + stats: python3 bench.py --rows 10000000 --steps 3 --warmup 2 --no-cpu-baseline --no-microbench --no-throughput --no-parity" > $OUT/summary.md
python3 tools/rocprof_summary.py stats $OUT/stats $OUT/summary.md
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/prof_r05_1e7/bench.json').read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'], 1), 'gpu kernel ms', round(d['gpu_kernel_ms_per_step'], 1), {n: round(v['frac'], 3) for n, v in d['kernels'].items() if v})
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
head -24 $OUT/summary.md
