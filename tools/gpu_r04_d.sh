#!/bin/bash
# round 4: k-means tests after the fused draw / update filter, KM fit timing, quick bench
set -u
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kpp or sskm or sklearn or lloyd or incremental or c1_shape or c4_shape or workspaces" > $O/km_tests2.txt 2>&1; rc=$?
tail -n 6 $O/km_tests2.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/km_fit_bench.py > $O/km_fit2.txt 2>&1; tail -n 5 $O/km_fit2.txt
bash tools/gpu_bench_quick.sh && cp gpurun_out/r03/bench_quick.json $O/bench_quick_sskm2.json
