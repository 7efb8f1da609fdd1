#!/bin/bash
# configs[3]: more 585-column tapes pending (accepted models waiting for G2) -- needs the longer ring of the variant library
set -o pipefail
out=gpurun_out/r05_cfg3c; rm -rf $out; mkdir -p $out
SO=$PWD/fokl_gpy_amd/csrc/variants/ring2048.so
for spec in "base=FOKL_X=1" "ring_448=FOKL_HIP_LIBRARY=$SO" "ring_900=FOKL_HIP_LIBRARY=$SO FOKL_PENDING_SEGMENTS=900" "ring_1400=FOKL_HIP_LIBRARY=$SO FOKL_PENDING_SEGMENTS=1400" "base2=FOKL_X=1" "ring_1400b=FOKL_HIP_LIBRARY=$SO FOKL_PENDING_SEGMENTS=1400"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs timeout -k 10 300 python3 bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-microbench --no-throughput > $out/cfg3_$name.json 2> $out/cfg3_$name.err || { tail -5 $out/cfg3_$name.err; exit 1; }
  python3 - $out/cfg3_$name.json $name <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1])
h = d['host_main_thread_s_per_step']
print(f"{sys.argv[2]:11s} ms {d['ms_per_step']:6.1f} parity {d['parity']['ok']} {d['parity'].get('max_draw_err_over_scale')} settle {1e3 * h['t_settle']:.0f} eigh {1e3 * h['t_eigh']:.0f} "
      f"tests {1e3 * h['phase_tests']:.0f} spectral cpu {h['pool_spectral_s']:.2f} submitted {h['spectral_submitted']:.0f} cpu {d['cpu_seconds_per_step']:.2f}")
PY
done
