#!/bin/bash
# round 6: the device chain's recursion (two sums off the critical path, one Newton step): parity tests + the headline fit
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_chain_device.py tests/test_config_goldens.py -x -q -m gpu > gpurun_out/r06_chain_tests.txt 2>&1 || { tail -40 gpurun_out/r06_chain_tests.txt; exit 1; }
tail -2 gpurun_out/r06_chain_tests.txt
bash tools/quick_bench.sh r06c FOKL_X=1 || exit 1
bash tools/quick_bench.sh r06c2 FOKL_X=1 || exit 1
python - <<'PY'
import json
for n in ('r06c', 'r06c2'):
    d = json.loads(open(f'gpurun_out/qb_{n}.json').read().strip().splitlines()[-1])
    h = d['host_main_thread_s_per_step']
    print(n, round(d['ms_per_step'], 2), 'chain kernel ms', d.get('device_chains', {}), {k: round(h[k]*1e3, 2) for k in ('phase_prepare','phase_model','phase_statistics','phase_tests','phase_wrap_up','t_final_verify','t_teardown') if k in h})
PY
