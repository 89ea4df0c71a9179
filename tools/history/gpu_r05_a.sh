#!/bin/bash
# round-5 first call: GPU suite of HEAD, then the in-situ question: per-kernel durations of the encoder with the LayerNorm folded into
# the GEMMs (default) and with separate LayerNorm kernels (SCD_LN_FUSE=0), and the stand-alone GEMMs on the same box
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/r05_gputest_head.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r05_gputest_head.txt
if [ $rc -ne 0 ]; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
stats() { # tag -- program args
  local tag=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof_$tag --output-format csv -- python3 "$@" > $out/prof_$tag.log 2>&1
  local rc=$?; echo "[rocprof $tag] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  local f=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && cp $f $out/r05_${tag}_kernel_stats.csv && head -n 8 $out/r05_${tag}_kernel_stats.csv | cut -c1-60,100-190
  rm -rf $out/prof_$tag
}
stats insitu_fused $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline || exit 1
export SCD_LN_FUSE=0
stats insitu_unfused $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline || exit 1
unset SCD_LN_FUSE
cd $R
timeout -k 10 300 python tools/gemm_bench.py 3990 > $out/r05_gemm_bench_3990.txt 2>&1; echo "[gemm_bench] rc=$?"; cat $out/r05_gemm_bench_3990.txt
