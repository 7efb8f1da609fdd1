#!/bin/bash
# configs[3] (585-column models): the eigen-update's product on the host (scipy's dgemm) / on the device, against the depth of
# derivation in wide sub-stages
set -o pipefail
out=gpurun_out/r05_cfg3b; rm -rf $out; mkdir -p $out
for spec in "host_w2=FOKL_EIGH_DGEMM_FROM=0" "dev_w2=FOKL_X=1" "dev_w4=FOKL_EIGH_UPDATE_DEPTH_WIDE=4" "dev_w8=FOKL_EIGH_UPDATE_DEPTH_WIDE=8" "dev_w16=FOKL_EIGH_UPDATE_DEPTH_WIDE=16" "host_w2b=FOKL_EIGH_DGEMM_FROM=0" "dev_w2b=FOKL_X=1"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs timeout -k 10 300 python3 bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-microbench --no-throughput > $out/cfg3_$name.json 2> $out/cfg3_$name.err || { tail -5 $out/cfg3_$name.err; exit 1; }
  python3 - $out/cfg3_$name.json $name <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1])
h = d['host_main_thread_s_per_step']
print(f"{sys.argv[2]:9s} ms {d['ms_per_step']:6.1f} parity {d['parity']['ok']} {d['parity'].get('max_draw_err_over_scale')} settle {1e3 * h['t_settle']:.0f} eigh {1e3 * h['t_eigh']:.0f} "
      f"spectral cpu {h['pool_spectral_s']:.2f} submitted {h['spectral_submitted']:.0f} updated {h['spectral_updated']:.0f} cpu {d['cpu_seconds_per_step']:.2f}")
PY
done
