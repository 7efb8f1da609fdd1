#!/bin/bash
# round 5: fits side by side on one MI355X against the box's CPU quota: P processes, 12 fits each
set -o pipefail
out=gpurun_out/r05l; mkdir -p $out
cat /sys/fs/cgroup/cpu.max; nproc
for side in 1 2 3 4 5; do
  FOKL_BENCH_SIDE_PROCS=$side timeout -k 10 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-microbench > $out/bench_$side.json 2> $out/bench_$side.err || { tail -5 $out/bench_$side.err; exit 1; }
  python - $side <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05l/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
tm = d['throughput_mode']; hc = tm['host_cpu']; w = tm['worker_s_per_fit']
print(f"{tm['procs']} processes: {tm['value']:9.0f} terms/s, {tm['fits_per_s']:5.1f} fits/s, {tm['ms_per_fit_per_process']:5.1f} ms per fit per process (alone: {d['ms_per_step']:.1f}); "
      f"CPUs used {hc['cpus_used']:.1f} of {hc['quota_cpus']:.0f}, periods throttled {hc['periods_throttled']} of {hc['periods']}; worker CPU-s per fit {w['cpu_s']:.3f}, "
      f"walker waiting for verdicts {1e3 * w['noise_verdict_wait_s']:.1f} ms, gpu kernels {d['gpu_kernel_ms_per_step']:.1f} ms per fit")
PY
done
