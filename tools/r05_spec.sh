#!/bin/bash
# throughput mode against the depth of the order book (tapes walked ahead of the decisions: wasted ones cost CPU of the quota)
out=gpurun_out/r05_spec; rm -rf $out; mkdir -p $out
for rep in 1 2; do
  for spec in 48 24 12 6; do
    FOKL_SPECULATION=$spec timeout -k 10 300 python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-microbench > $out/s${spec}_$rep.json 2> $out/s${spec}_$rep.err || { tail -5 $out/s${spec}_$rep.err; exit 1; }
    python3 - $out/s${spec}_$rep.json $spec <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tm = d['throughput_mode']; h = d['host_main_thread_s_per_step']
print(f"speculation {sys.argv[2]:>3s}: alone {d['ms_per_step']:.2f} ms, cpu-s {d['cpu_seconds_per_step']:.3f}, rewound {h['tapes_rewound']:.0f} wasted {h['tapes_wasted']:.0f}; "
      f"{tm['procs']} processes {tm['value']:.0f} terms/s, {tm['ms_per_fit_per_process']:.1f} ms per fit each, cpus {tm['host_cpu']['cpus_used']:.1f}, "
      f"throttled {tm['host_cpu']['periods_throttled']}/{tm['host_cpu']['periods']}, worker cpu-s {tm['worker_s_per_fit']['cpu_s']:.3f}")
PY
  done
done
