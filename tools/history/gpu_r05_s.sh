#!/bin/bash
# Does attention run faster when q / k / v of the launch fit the 256-MB Infinity Cache?  rocprofv3 kernel stats of encodes at 128 images
# (q / k / v 116 MB) with non-temporal and with plain C stores of the QKV GEMM, against the default 3,990-image launch, same box.
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
o=$R/gpurun_out/r05/r05_attn_mall_probe.txt; : > $o
for cfg in "128 -1" "128 0" "256 -1" "256 0" "3990 -1"; do
  set -- $cfg; b=$1; nt=$2
  if [ "$nt" = "-1" ]; then unset SCD_GEMM_NT; else export SCD_GEMM_NT=$nt; fi
  echo "== B=$b SCD_GEMM_NT=${SCD_GEMM_NT:-by-size}" >> $o
  bash $R/tools/gpu_prof_any.sh attn_${b}_${nt} tools/attn_bench.py $b >> $o 2>&1 || exit 1
done
grep -E "^==|attention_persist|ELi8ELi0ELb1ELb0ELi1" $o
