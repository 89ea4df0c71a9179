#!/bin/bash
# ring-fill schedule experiment (-DW4_WSPLIT): correctness of the variant build, then a same-box A/B of whole bench runs
set -u
mkdir -p gpurun_out/r05
SCD_HIP_LIB=$PWD/scd_amd/lib/libscd_hip_wsplit4.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or tower or encoder_large" 2>&1 | tail -n 3
bash tools/gpu_r04_abn.sh scd_amd/lib/libscd_hip_wsplit4.so scd_amd/lib/libscd_hip_wsplit2.so 2>&1 | tee gpurun_out/r05/r05_wsplit_ab.txt
