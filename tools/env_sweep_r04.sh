#!/bin/bash
# tools/env_sweep_r04.sh: the benchmark fit under a few thread plans / look-ahead depths (development aid)
out=gpurun_out/env_sweep_r04.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-microbench --no-throughput 2>/dev/null | python -c "
import sys, json
p = json.loads(sys.stdin.read().strip().split('\n')[-1])
h = p['host_main_thread_s_per_step']
keys = ['t_eigh','t_chain','t_kill_loop','pool_noise_s','pool_finish_s','pool_spectral_s','pool_chain_s','noise_verdict_wait_s','noise_queue_wait_s','t_final_verify','tapes_rewound','tapes_wasted','pool_bulk_s','walker_wait_s','t_resid','t_final_draws','tapes_materialised','path_repredicted','spectral_submitted','phase_prepare','phase_model','phase_statistics','phase_tests','phase_wrap_up','t_pool_up','t_teardown','t_search_body','seconds']
print('ms_per_step %.2f cpu_s %.3f parity %s' % (p['ms_per_step'], p['cpu_seconds_per_step'], p['parity']['ok']), {k: (round(h[k],4) if isinstance(h.get(k), float) else h.get(k)) for k in keys})
" >> $out 2>&1
}
run FOKL_X=0
run FOKL_X=0
run FOKL_SPECTRAL_THREADS=9
run FOKL_SPECTRAL_THREADS=10 FOKL_BULK_THREADS=3
cat $out
