#!/bin/bash
# rocprofv3 kernel stats of one C4-size bench step (sim / k-means / vote kernels at N = 160,146, K = 1000)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03/p_c4; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 $R/bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1
echo "[rocprof bench c4] rc=$?"
f=$(find $out -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && cp $f $R/gpurun_out/r03/r03_bench_c4_kernel_stats.csv && grep -i "sim_\|wmax\|estep\|muf_\|mstep\|finalize\|vote\|kpp" $R/gpurun_out/r03/r03_bench_c4_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-170
rm -rf $out
