#!/bin/bash
# timing ablations of the similarity kernels: tools/gpu_sim_x.sh "X values" [k]
set -u
out=gpurun_out/sim_x; mkdir -p $out
export SCD_SIM_RB=${SCD_SIM_RB:-8}
for x in $1; do
  SCD_SIM_X=$x timeout -k 10 120 python tools/sim_bench.py 126976 ${2:-3} > $out/x$x.log 2>&1; rc=$?
  echo "[X=$x] rc=$rc"; grep sim_topk $out/x$x.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
