#!/bin/bash
OUT=gpurun_out/r05_clock2; rm -rf $OUT; mkdir -p $OUT
SO=$PWD/fokl_gpy_amd/csrc/variants/stamp.so
S=${K2_SHAPES:-56x75,56x101,56x128,56x176}
FOKL_HIP_LIBRARY=$SO K2_N=1000000 K2_SHAPES=$S timeout -k 10 300 python3 tools/k2_clock.py 3000 > $OUT/clock_1e6.txt 2>&1 || { tail $OUT/clock_1e6.txt; exit 1; }
FOKL_HIP_LIBRARY=$SO K2_N=20000000 K2_SHAPES=$S timeout -k 10 400 python3 tools/k2_clock.py 300 > $OUT/clock_2e7.txt 2>&1 || exit 1
cat $OUT/clock_1e6.txt $OUT/clock_2e7.txt
