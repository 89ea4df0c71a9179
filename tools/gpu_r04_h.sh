#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
echo "== library WITHOUT the wait states behind the asm stores (expected to fail)"
SCD_HIP_LIB=$PWD/scd_amd/lib/libscd_hip_nonop.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "large_launch" 2>&1 | tail -n 4
echo "== shipped library"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_launch or batch_invariance" 2>&1 | tail -n 3
