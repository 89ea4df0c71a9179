#!/bin/bash
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
for i in 1 2 3; do
SCD_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 1 --warmup 1 --images 1200 --n-cluster 12 --vocab 1024 --batch 665 --no-cpu-baseline > $out/l4.out 2> $out/l4.err; echo "4 ranks, 4 streams, try $i: rc=$?"
done
grep -n "terminate\|what()" $out/l4.err | head
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_ranks or two_ranks or multi_rank or sharded_loops or bench_config_c3" > $out/r05_tests_m.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r05_tests_m.txt
