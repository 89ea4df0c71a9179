#!/bin/bash
# SCD_GEMM_NG sweep: tile columns per n-group, at 665 / 1,995 / 3,990 images per launch
set -u
O=gpurun_out/r04; mkdir -p $O
: > $O/gemm_ng_sweep2.txt
for b in 665 1995 3990; do
for ng in 0 3 5 9 12; do
  echo "== batch $b SCD_GEMM_NG=$ng" >> $O/gemm_ng_sweep2.txt
  SCD_GEMM_NG=$ng timeout -k 10 200 python tools/gemm_bench.py $b 2>&1 | grep -v amdgpu.ids | head -4 | awk 'NR==1||NR==3' >> $O/gemm_ng_sweep2.txt || exit 1
done
done
cat $O/gemm_ng_sweep2.txt
