#!/bin/bash
# In-kernel clock of the Gram kernel: a diagnostic build of the library (-DFOKL_GT_STAMP: s_memtime / s_memrealtime stamps
# around every workgroup's loop) + tools/k2_clock.py.  Run on the GPU box from the repo root.
set -e
cd fokl_gpy_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DFOKL_GT_STAMP -c -o /tmp/fokl_hip_stamp.o fokl_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o /tmp/libfokl_hip_stamp.so /tmp/fokl_hip_stamp.o fokl_sampler.o fokl_sampler_wide.o fokl_vlog.o fokl_hostpool.o fokl_integrate.o fokl_stream.o fokl_search.o fokl_clean.o -ldl -lpthread -lmvec -lm
cd ../..
FOKL_HIP_LIBRARY=/tmp/libfokl_hip_stamp.so python3 tools/k2_clock.py "$@"
