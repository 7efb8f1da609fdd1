#!/bin/bash
# round 6: baseline of the tree -- config goldens on the GPU, the headline bench, the walker / driver timeline of a fit
set -o pipefail
out=gpurun_out/r06_base; rm -rf $out; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_config_goldens.py -x -q -m gpu > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench --no-throughput > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
FOKL_POOL_TRACE=$out/trace.txt timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-microbench --no-throughput --no-parity > $out/bench_traced.json 2> $out/bench_traced.err || { tail -20 $out/bench_traced.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_base/bench.json').read().strip().splitlines()[-1])
h = d.get('host_main_thread_s_per_step', {})
print('ms', round(d['ms_per_step'], 2), 'parity', d.get('parity', {}).get('ok'),
      {k: round(h[k] * 1e3, 2) for k in ('phase_prepare', 'phase_model', 'phase_statistics', 'phase_tests', 'phase_wrap_up', 't_final_verify', 't_teardown', 't_eigh', 't_kill_loop', 'pool_noise_s', 'noise_verdict_wait_s', 'noise_queue_wait_s', 'pool_spectral_s', 't_settle') if k in h},
      d.get('kill_decisions'), 'cpu', d.get('cpu_seconds_per_step'))
PY
python tools/fit_timeline.py $out/trace.txt --fit -2 > $out/timeline.txt 2>&1
head -5 $out/timeline.txt
