#!/bin/bash
# round 5, first GPU call: GPU tests, the headline bench with the kill tests decided directly and from G2 (A/B), box facts
set -o pipefail
mkdir -p gpurun_out/r05a
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)))"; lscpu | grep -i "model name\|^CPU(s)\|Thread\|L3" ) > gpurun_out/r05a/box.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/r05a/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r05a/gpu_tests.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench_direct.json 2> gpurun_out/r05a/bench_direct.err || { tail -20 gpurun_out/r05a/bench_direct.err; exit 1; }
FOKL_KILL_DECIDE=g2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench --no-throughput > gpurun_out/r05a/bench_g2.json 2> gpurun_out/r05a/bench_g2.err || { tail -20 gpurun_out/r05a/bench_g2.err; exit 1; }
python - <<'PY'
import json
for name in ('direct', 'g2'):
    d = json.loads(open(f'gpurun_out/r05a/bench_{name}.json').read().strip().splitlines()[-1])
    h = d.get('host_main_thread_s_per_step', {})
    print(name, 'ms', round(d['ms_per_step'], 2), 'parity', d.get('parity', {}).get('ok'), d.get('parity', {}).get('max_draw_err_over_scale'),
          {k: round(h[k] * 1e3, 2) for k in ('phase_prepare', 'phase_model', 'phase_statistics', 'phase_tests', 'phase_wrap_up', 't_final_verify', 't_teardown', 't_eigh', 't_kill_loop', 'pool_noise_s', 'noise_verdict_wait_s', 'noise_queue_wait_s', 'pool_spectral_s', 't_settle') if k in h},
          {k: h.get(k) for k in ('direct_tests', 'direct_max_rel', 'chains_cancelled', 'device_chains', 'spectral_submitted', 'spectral_updated', 'guessed', 'guess_waits')},
          'tp', d.get('throughput_mode', {}) and d['throughput_mode'].get('value'))
PY
