#!/bin/bash
# finish / chain thread counts against the statistics phase (the sub-stage model's own chain): interleaved A/B on one box
out=gpurun_out/r05_threads; rm -rf $out; mkdir -p $out
for rep in 1 2; do
  for plan in default f3 f4 f4c3 f6; do
    case $plan in default) E="";; f3) E="FOKL_FINISH_THREADS=3";; f4) E="FOKL_FINISH_THREADS=4";; f4c3) E="FOKL_FINISH_THREADS=4 FOKL_CHAIN_THREADS=3";; f6) E="FOKL_FINISH_THREADS=6";; esac
    env $E timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-microbench --no-throughput > $out/${plan}_$rep.json 2> $out/${plan}_$rep.err || { tail -5 $out/${plan}_$rep.err; exit 1; }
    python3 - $out/${plan}_$rep.json $plan <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d['host_main_thread_s_per_step']
print(f"{sys.argv[2]:8s} ms {d['ms_per_step']:.2f}  " + ' '.join(f"{k[6:]} {1e3 * h[k]:.2f}" for k in h if k.startswith('phase_')) +
      f"  final_verify {1e3 * h['t_final_verify']:.2f} cpu {d['cpu_seconds_per_step']:.3f} parity {d['parity']['ok']}")
PY
  done
done
