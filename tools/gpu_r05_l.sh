#!/bin/bash
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
SCD_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 1 --warmup 1 --images 1200 --n-cluster 12 --vocab 1024 --batch 665 --no-cpu-baseline > $out/l4.out 2> $out/l4.err; echo "rc=$?"
grep -v "Gloo\] Rank" $out/l4.err | grep -v Warning | tail -n 40
echo ---- lockstep off
SCD_LLOYD_LOCKSTEP=0 SCD_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 1 --warmup 1 --images 1200 --n-cluster 12 --vocab 1024 --batch 665 --no-cpu-baseline > $out/l4b.out 2> $out/l4b.err; echo "rc=$?"
echo ---- streams 1
SCD_LLOYD_STREAMS=1 SCD_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 1 --warmup 1 --images 1200 --n-cluster 12 --vocab 1024 --batch 665 --no-cpu-baseline > $out/l4c.out 2> $out/l4c.err; echo "rc=$?"
grep -v "Gloo\] Rank" $out/l4c.err | grep -v Warning | tail -n 15
