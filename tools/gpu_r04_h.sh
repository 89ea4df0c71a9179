#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip_towers or encoder or dino or gemm or extract_feature or text_tower" > $O/enc_tests.txt 2>&1; rc=$?
tail -n 3 $O/enc_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/gemm_bench.py 3990 2>&1 | grep -v amdgpu.ids | head -4
bash tools/gpu_bench_quick.sh --steps 2 --warmup 1 2>&1 | grep -e "^[0-9]"
