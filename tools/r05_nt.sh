#!/bin/bash
# round 5: non-temporal column loads in K2 (LDS-DMA aux = 2) / K3 keep the inputs in the Infinity Cache for K1: the default
# build (FOKL_STREAM_NT=1, no touch launch) against the round-4 arrangement (default-policy loads, with and without the touch)
set -o pipefail
out=gpurun_out/r05e
mkdir -p $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench --no-throughput > $out/bench_$name.json 2> $out/bench_$name.err || { tail -20 $out/bench_$name.err; exit 1; }
  python - $name <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05e/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
k = d['kernels']
def f(name, *keys):
    e = k.get(name) or {}
    return ' '.join(f"{key}={e.get(key):.3f}" if isinstance(e.get(key), float) else f"{key}={e.get(key)}" for key in keys)
print(f"{sys.argv[1]:14s} ms {d['ms_per_step']:.2f} parity {d['parity']['ok']} gpu_ms {d['gpu_kernel_ms_per_step']:.2f} | K1 {f('basis_build','frac','avg_ms','kernel_only_frac')} | touch {f('inputs_touch','launches','avg_ms')} | gram {f('gram','frac','avg_ms')} | mfma {f('gram_mfma','frac','avg_ms')} | resid {f('resid','frac','avg_ms')} | resid_mf {f('resid_matrix_free','frac','avg_ms')}")
PY
}
PLAIN=$PWD/fokl_gpy_amd/csrc/variants/plain.so
run nt FOKL_X=1
run plain_touch FOKL_K1_TOUCH=1 FOKL_HIP_LIBRARY=$PLAIN
run nt_again FOKL_X=1
run plain_notouch FOKL_HIP_LIBRARY=$PLAIN
run nt_touch FOKL_K1_TOUCH=1
run nt_third FOKL_X=1
