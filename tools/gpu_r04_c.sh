#!/bin/bash
# round 4: full GPU suite, then quick bench (SSKM default and --cluster KM)
set -u
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gputest.txt 2>&1; rc=$?
tail -n 8 $O/gputest.txt
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_bench_quick.sh && cp gpurun_out/r03/bench_quick.json $O/bench_quick_sskm.json
bash tools/gpu_bench_quick.sh --cluster KM && cp gpurun_out/r03/bench_quick.json $O/bench_quick_km.json
