#!/bin/bash
# same-box A/B of two builds of the library (FOKL_HIP_LIBRARY): tools/r06_ab.sh other.so [rounds]
set -o pipefail
other=$1; rounds=${2:-3}
for r in $(seq 1 $rounds); do
  bash tools/quick_bench.sh ab_new_$r FOKL_X=1 || exit 1
  bash tools/quick_bench.sh ab_old_$r FOKL_HIP_LIBRARY=$PWD/$other || exit 1
done
