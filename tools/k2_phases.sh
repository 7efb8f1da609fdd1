#!/bin/bash
# Where a workgroup of gram_tiles_dma_kernel spends its cycles: a diagnostic build of the library (-DFOKL_GD_STAMP: s_memtime
# stamps between the phases of every chunk -- issue of the next chunk's LDS-DMA pieces, the MFMA loop, the wait for the
# wavefront's own pieces, the barrier) + tools/k2_phases.py.  Run on the GPU box from the repo root.
set -e
cd fokl_gpy_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DFOKL_GD_STAMP -c -o /tmp/fokl_hip_stamp.o fokl_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o /tmp/libfokl_hip_stamp.so /tmp/fokl_hip_stamp.o fokl_sampler.o fokl_sampler_wide.o fokl_vlog.o fokl_hostpool.o fokl_integrate.o -ldl -lpthread -lmvec -lm
cd ../..
FOKL_HIP_LIBRARY=/tmp/libfokl_hip_stamp.so python3 tools/k2_phases.py "$@"
