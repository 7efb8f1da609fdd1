#!/bin/bash
out=gpurun_out/r05_k3q; rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k3 or resid" > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
timeout -k 10 200 python3 tools/k3_probe.py 1000000 > $out/k3_1e6.txt 2>&1 || { tail $out/k3_1e6.txt; exit 1; }
timeout -k 10 300 python3 tools/k3_probe.py 10000000 > $out/k3_1e7.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/k3_probe.py 1000000 0 > $out/k3_1e6_splines.txt 2>&1 || { tail $out/k3_1e6_splines.txt; exit 1; }
cat $out/k3_1e6.txt $out/k3_1e7.txt $out/k3_1e6_splines.txt
timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-microbench --no-throughput > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_k3q/bench.json').read().strip().splitlines()[-1])
k = d['kernels']
print('ms_per_step', round(d['ms_per_step'], 2), 'parity', d['parity']['ok'], {n: round(v['frac'], 3) for n, v in k.items() if v})
print(k['resid_matrix_free'])
PY
