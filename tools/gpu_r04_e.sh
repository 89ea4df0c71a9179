#!/bin/bash
set -u
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sklearn_102 or sklearn_kmeans_c" > $O/km_tests3.txt 2>&1; rc=$?
tail -n 3 $O/km_tests3.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/km_fit_bench.py 2>&1 | grep -v amdgpu.ids
bash tools/gpu_prof_km.sh 2>&1 | grep -e muf_filter -e kg_exact
