#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip_towers or encoder or dino or gemm or extract_feature or text_tower" > $O/enc_tests.txt 2>&1; rc=$?
tail -n 3 $O/enc_tests.txt
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_r04_ab.sh scd_amd/lib/libscd_hip_prev.so
