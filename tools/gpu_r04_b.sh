#!/bin/bash
# round 4: the new --cluster KM tests, then a KM fit timing at C2 size
set -u
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sklearn or c1_shape or incremental_mstep or lloyd_run" > $O/km_tests.txt 2>&1; rc=$?
tail -n 25 $O/km_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/km_fit_bench.py > $O/km_fit.txt 2>&1; rc=$?
tail -n 20 $O/km_fit.txt
exit $rc
