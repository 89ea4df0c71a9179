#!/bin/bash
# round-3 ablations of the similarity kernels on ONE box, written to gpurun_out/r03/r03_sim_ablations.txt
# (needs the -DSCD_ABLATE build: python -m scd_amd.build --ablate).  SCD_SIM_X bits (timing only, results are wrong):
#   1 no epilogue, 2 no ring fills, 64 no second-key path (instantiated: 1, 3, 64)
set -u
mkdir -p gpurun_out/r03
o=gpurun_out/r03/r03_sim_ablations.txt
{
echo "# tools/gpu_r03_sim_ablations.sh: scd_sim_topk CALL (wmax + kernel + refine + exact-pass launches) on 126976 x 21000 x 512, k = 3,"
echo "# random unit vectors, HIP events over 5 calls; SCD_SIM_RB = 8 (sim_topk_rb8_kernel, default), 16 (sim_topk_rc_kernel), 1 (round 2's four-wave sim_topk_rb_kernel)"
echo "# SCD_SIM_X bits: 1 no epilogue, 2 no ring fills, 64 no second-key path; X != 0 lines are the kernel alone, X = 0 lines the whole call (refine = ~0.1 ms of every line)"
} > $o
RBS="8 16" bash tools/gpu_sim_x.sh "0 3 1 64" 3 >> $o 2>&1 || exit 1
RBS="1" bash tools/gpu_sim_x.sh "0 3" 3 >> $o 2>&1 || exit 1
cat $o
