#!/bin/bash
# tools/quick_bench.sh NAME [ENV=VALUE ...]: a short timed run of the headline fit without side measurements; prints
# value, ms per step, CPU-seconds and the host breakdown fields that matter (development aid, GPU box).  QB_ARGS replaces
# "--steps 8 --warmup 3" (e.g. QB_ARGS="--config 3 --steps 3 --warmup 1").
name=$1; shift
mkdir -p gpurun_out
env "$@" python bench.py ${QB_ARGS:---steps 8 --warmup 3} --no-cpu-baseline --no-microbench --no-throughput > gpurun_out/qb_$name.json 2> gpurun_out/qb_$name.err || { tail -c 400 gpurun_out/qb_$name.err; exit 1; }
python - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host_main_thread_s_per_step"]
keys = ('spectral_updated', 'spectral_device', 't_eigh', 'pool_spectral_s', 'pool_noise_s', 'noise_verdict_wait_s', 'phase_tests', 'phase_model', 'phase_statistics', 't_kill_loop', 'spectral_submitted', 'tapes_rewound')
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 2), round(d["cpu_seconds_per_step"], 3), d["parity"].get("ok"),
      d["parity"].get("max_draw_err_over_scale"), {k: round(h.get(k, 0), 4) for k in keys})
PY
