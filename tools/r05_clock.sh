#!/bin/bash
# In-kernel clock of the Gram launches at N = 1e6 (0.2-0.3 ms launches) and N = 2e7 (3-5 ms launches) after seconds of
# back-to-back launches (MI355X_MICROARCH.md, DVFS item 6), on the stamped diagnostic build
# (tools/k2_variants.sh build stamp:"-DFOKL_GT_STAMP" on the CPU box first); then K3 stored columns against matrix-free.
OUT=gpurun_out/r05_clock; rm -rf $OUT; mkdir -p $OUT
SO=$PWD/fokl_gpy_amd/csrc/variants/stamp.so
S=${K2_SHAPES:-56x75,56x101,56x128,56x176}
FOKL_HIP_LIBRARY=$SO K2_N=1000000 K2_SHAPES=$S timeout -k 10 300 python3 tools/k2_clock.py 6000 > $OUT/clock_1e6.txt 2>&1 || exit 1
FOKL_HIP_LIBRARY=$SO K2_N=20000000 K2_SHAPES=$S timeout -k 10 400 python3 tools/k2_clock.py 600 > $OUT/clock_2e7.txt 2>&1 || exit 1
cat $OUT/clock_1e6.txt $OUT/clock_2e7.txt
timeout -k 10 200 python3 tools/k3_probe.py 1000000 > $OUT/k3_1e6.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/k3_probe.py 10000000 > $OUT/k3_1e7.txt 2>&1 || exit 1
cat $OUT/k3_1e6.txt $OUT/k3_1e7.txt
for rep in 1 2; do
  for k3 in matrixfree columns; do
    FOKL_K3=$k3 timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-microbench --no-throughput > $OUT/bench_${k3}_$rep.json 2> $OUT/bench_${k3}_$rep.err || exit 1
    python3 - $OUT/bench_${k3}_$rep.json $k3 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d['kernels']
print(sys.argv[2], 'ms_per_step', round(d['ms_per_step'], 2), 'gpu kernel ms', round(d['gpu_kernel_ms_per_step'], 2),
      'resid', round(k['resid']['total_ms'] / d['steps'], 3), 'ms in', k['resid']['launches'] // d['steps'], 'launches;  matrix-free',
      round((k['resid_matrix_free'] or {}).get('total_ms', 0) / d['steps'], 3), 'ms in', (k['resid_matrix_free'] or {}).get('launches', 0) // d['steps'])
PY
  done
done
