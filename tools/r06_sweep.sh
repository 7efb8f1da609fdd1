#!/bin/bash
# same-box sweep of one environment setting: tools/r06_sweep.sh NAME v1 v2 ... (each value twice, interleaved)
set -o pipefail
name=$1; shift
for round in 1 2; do
  for v in "$@"; do
    bash tools/quick_bench.sh sw_${name}_${v}_$round $name=$v | cut -c1-60 || exit 1
  done
done
