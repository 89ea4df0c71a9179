#!/bin/bash
# round-3 evidence for the similarity path: full GPU suite, call-level bench incl. hipBLASLt on the same tensors, rocprofv3 kernel stats
set -u
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/gputest.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 5 $out/gputest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for k in 3 5 1; do timeout -k 10 120 python tools/sim_bench.py 126976 $k hipblaslt; done > $out/sim_bench.txt 2>&1
grep -E "sim_topk|torch.mm" $out/sim_bench.txt
bash tools/gpu_sim_stats.sh 3 r03
