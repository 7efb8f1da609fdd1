#!/bin/bash
# same-box A/B of an environment setting against the default: tools/r06_ab_env.sh NAME=VALUE [rounds]
set -o pipefail
setting=$1; rounds=${2:-3}
for r in $(seq 1 $rounds); do
  bash tools/quick_bench.sh ab_default_$r FOKL_X=1 || exit 1
  bash tools/quick_bench.sh ab_other_$r $setting || exit 1
done
