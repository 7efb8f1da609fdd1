#!/bin/bash
# building the coming sub-stage ahead (K1 + K2 against every column that can still be in the model) where the device binds:
# off beyond 5e6 rows since round 3, when the host took 100+ ms per fit -- measured again with round 5's host
out=gpurun_out/r05_foresight; rm -rf $out; mkdir -p $out
for f in default 8 0 8 0; do
  if [ $f = default ]; then E=""; else E="FOKL_FORESIGHT=$f"; fi
  env $E N_SCALING_REPS=3 N_SCALING_WARMUP=2 timeout -k 10 500 python3 tools/n_scaling.py 5e6 1e7 2e7 > $out/f_$f.txt 2>&1 || { tail $out/f_$f.txt; exit 1; }
  echo "== FOKL_FORESIGHT=$f"; cut -c1-200 $out/f_$f.txt
done
