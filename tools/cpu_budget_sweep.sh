#!/bin/bash
# CPU-budget sweep on a 1-GPU box: the benchmark fit with the process confined to 2 / 3 / 4 / 6 / 8 / 16 logical CPUs of
# one L3 domain (the thread plan of engine._thread_plan follows the affinity mask).  Predicts what a rank gets out of its
# share of a node's CPU quota when 8 ranks run side by side.
out=${1:-gpurun_out/r2_cpu_sweep.txt}
: > $out
for c in 2 3 4 6 8 16; do
  FOKL_BENCH_PIN=0 taskset -c 0-$((c-1)) python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-microbench --no-parity --no-throughput > /tmp/sweep_$c.json 2>/dev/null
  python - <<PY >> $out
import json
d=json.load(open('/tmp/sweep_$c.json')); h=d['host_main_thread_s_per_step']
cpu=h['pool_noise_s']+h['pool_chain_s']+h['pool_finish_s']+h['pool_spectral_s']
print('cpus $c  ms/fit %.1f  terms/s %.0f  process CPU-s/fit %.3f  pool busy-s/fit %.3f (noise %.3f chain %.3f finish %.3f spectral %.3f)  chains on %s' % (d['ms_per_step'], d['value'], d.get('cpu_seconds_per_step', float('nan')), cpu, h['pool_noise_s'], h['pool_chain_s'], h['pool_finish_s'], h['pool_spectral_s'], 'device' if h.get('device_chains') else 'host'))
PY
done
cat $out
