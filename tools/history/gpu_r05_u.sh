#!/bin/bash
# attention kernel duration with and without the T = 197 specialisation (rocprofv3 kernel stats, 3,990-image encodes, one box)
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
o=$R/gpurun_out/r05/r05_attn_t197.txt; : > $o
for m in 0 1 0 1; do
  export SCD_ATTN_T197=$m
  echo "== SCD_ATTN_T197=$m" >> $o
  bash $R/tools/gpu_prof_any.sh t197_$m tools/attn_bench.py 3990 >> $o 2>&1 || exit 1
done
grep -E "^==|attention_persist|encode B" $o
