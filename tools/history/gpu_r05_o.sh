#!/bin/bash
# how long does a tile's FIRST chunk wait at its barrier (behind the previous tile's stores)?  -DSCD_ABLATE cycle counters
set -u
export SCD_HIP_LIB=$PWD/scd_amd/lib/libscd_hip_ablate.so
for x in 192 2240; do
  echo "== SCD_GEMM_X=$x"
  SCD_GEMM_X=$x timeout -k 10 300 python tools/gemm_bench.py 3990 2>&1 | grep -E "w4 m=|TFLOP" | head -20
done
