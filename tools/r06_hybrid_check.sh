#!/bin/bash
# round 6: the sharded configs[3] fits of the rehearsal (two ranks on one GPU over TCP) against the d250 golden, one setting at a time
export FOKL_BENCH_SHARE_GPU=1 FOKL_BENCH_SHARDED_OVER_TCP=1
common="--no-cpu-baseline --no-microbench --no-throughput"
i=0
for setting in "$@"; do
  i=$((i + 1))
  mode=${setting%%:*}; envs=${setting#*:}
  env $envs timeout -k 10 500 python bench.py --gpus 2 --config 3 --mode $mode --steps 1 --warmup 0 $common > gpurun_out/hyb_$i.json 2> gpurun_out/hyb_$i.err
  echo "== $setting rc $?"
  grep -o "PARITY MISMATCH[^{]*{[^}]*}" gpurun_out/hyb_$i.err | cut -c1-330
  python - $i <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/hyb_{sys.argv[1]}.json").read().strip().splitlines() if l.startswith('{')][-1])
    p = d.get('parity') or {}
    print('   parity', p.get('ok'), p.get('max_rel_bic'), p.get('max_draw_err_over_scale'), 'every rank', p.get('ok_on_every_rank'), 'ms', round(d['ms_per_step'], 1))
except Exception as exc:
    print('   no line', exc)
PY
done
