#!/bin/bash
out=gpurun_out/r05_head; rm -rf $out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_config_goldens.py tests/test_gpu_parity.py -x -q -m gpu > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
for rep in 1 2 3; do
  for h in 1 0; do
    FOKL_HEAD_START=$h timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-microbench --no-throughput > $out/h${h}_$rep.json 2> $out/h${h}_$rep.err || { tail -5 $out/h${h}_$rep.err; exit 1; }
    python3 - $out/h${h}_$rep.json $h <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d['host_main_thread_s_per_step']
print(f"head start {sys.argv[2]}: {d['ms_per_step']:.2f} ms  parity {d['parity']['ok']}  " + ' '.join(f"{k[6:]} {1e3 * h[k]:.2f}" for k in h if k.startswith('phase_')) + f" pool_up {1e3 * h['t_pool_up']:.2f}")
PY
  done
done
