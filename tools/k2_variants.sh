#!/bin/bash
# Development aid: builds libfokl_hip.so variants with extra -D flags (cross-compiles here; the .so files travel to the GPU
# box) and, with `run`, times the benchmark fit's Gram shapes back to back on each.
#   tools/k2_variants.sh build name1:"-DFOO=1 -DBAR=2" name2:"..."     (on the CPU box)
#   tools/k2_variants.sh run [reps] [K2_SHAPES]                         (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VAR=$ROOT/fokl_gpy_amd/csrc/variants
if [ "$1" = build ]; then
  shift
  mkdir -p $VAR
  cd $ROOT/fokl_gpy_amd/csrc
  make -s                                  # host objects
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $flags -c -o $VAR/$name.o fokl_hip.hip &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $VAR/$name.so $VAR/$name.o fokl_sampler.o fokl_sampler_wide.o fokl_vlog.o fokl_hostpool.o fokl_integrate.o fokl_stream.o fokl_search.o fokl_clean.o -ldl -lpthread -lmvec -lm &&
      rm -f $VAR/$name.o && echo "built $name ($flags)" ) &
  done
  wait
else
  shift || true
  reps=${1:-20}
  for so in $VAR/*.so; do
    echo "== $(basename $so .so)"
    FOKL_HIP_LIBRARY=$so K2_SHAPES=${2:-56x80,56x98,56x128,56x142,56x176,28x120,8x100} timeout -k 10 300 python3 $ROOT/tools/k2_experiment.py $reps 2
  done
fi
