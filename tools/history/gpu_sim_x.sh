#!/bin/bash
# timing ablations of the similarity kernels, both row-block kernels on the same box: tools/gpu_sim_x.sh "X values" [k]
set -u
[ -f scd_amd/lib/libscd_hip_ablate.so ] && export SCD_HIP_LIB=$PWD/scd_amd/lib/libscd_hip_ablate.so   # SCD_SIM_X needs the -DSCD_ABLATE build (python -m scd_amd.build --ablate)
out=gpurun_out/sim_x; mkdir -p $out
for x in $1; do
  for rb in ${RBS:-8 16}; do
    SCD_SIM_RB=$rb SCD_SIM_X=$x timeout -k 10 120 python tools/sim_bench.py 126976 ${2:-3} > $out/rb${rb}_x$x.log 2>&1; rc=$?
    echo "[RB=$rb X=$x] rc=$rc $(grep sim_topk $out/rb${rb}_x$x.log | sed 's/n=126976 v=21000: //; s/fallback.*//' | tr '\n' ' ')"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
done
