#!/bin/bash
# the headline fit against the depth of the order book, interleaved
out=gpurun_out/r05_spec2; rm -rf $out; mkdir -p $out
for rep in 1 2 3; do
  for spec in 48 32 24 16; do
    FOKL_SPECULATION=$spec timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-microbench --no-throughput > $out/s${spec}_$rep.json 2> $out/s${spec}_$rep.err || { tail -5 $out/s${spec}_$rep.err; exit 1; }
    python3 - $out/s${spec}_$rep.json $spec <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d['host_main_thread_s_per_step']
print(f"speculation {sys.argv[2]:>3s}: {d['ms_per_step']:.2f} ms, cpu-s {d['cpu_seconds_per_step']:.3f}, rewound {h['tapes_rewound']:.0f} wasted {h['tapes_wasted']:.0f}, "
      f"walker busy {1e3 * h['pool_noise_s']:.1f} verdict wait {1e3 * h['noise_verdict_wait_s']:.1f} queue wait {1e3 * h['noise_queue_wait_s']:.1f}; "
      + ' '.join(f"{k[6:]} {1e3 * h[k]:.2f}" for k in h if k.startswith('phase_')))
PY
  done
done
