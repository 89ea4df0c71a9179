#!/bin/bash
set -u
mkdir -p gpurun_out/r05
SCD_HIP_LIB=$PWD/scd_amd/lib/libscd_hip_late4.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or tower or encoder_large" 2>&1 | tail -n 2
bash tools/gpu_r04_abn.sh scd_amd/lib/libscd_hip_late3.so scd_amd/lib/libscd_hip_late4.so 2>&1 | tee gpurun_out/r05/r05_late_tm_ab.txt
