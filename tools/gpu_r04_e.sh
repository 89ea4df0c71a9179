#!/bin/bash
set -u
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kpp or sklearn_102" > $O/km_tests3.txt 2>&1; rc=$?
tail -n 3 $O/km_tests3.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/km_fit_bench.py > $O/km_fit3.txt 2>&1; tail -n 4 $O/km_fit3.txt
bash tools/gpu_bench_quick.sh --steps 1 --warmup 1 2>&1 | grep -e "^[0-9]" -e "round_us" -e fit_wall
