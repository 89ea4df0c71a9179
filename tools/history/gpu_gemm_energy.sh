#!/bin/bash
# energy attribution of the ViT GEMMs by timing ablations under the power cap -> gpurun_out/r04/r04_gemm_energy.txt
set -u
O=gpurun_out/r04; mkdir -p $O
export SCD_HIP_LIB=$PWD/scd_amd/lib/libscd_hip_ablate.so
{ echo "# tools/gpu_gemm_energy.sh: tools/gemm_energy.py under SCD_GEMM_X of the -DSCD_ABLATE build (2 no stores, 4 every tile loads the same L2-resident panels, 16 no epilogue), 3,990 images per launch, random operands, 2.5 s per line after 1 s of warm-up"
for shape in "786432 3072 768 1 0" "786432 2304 768 0 0" "786432 768 3072 0 1"; do
  for x in 0 2 4 6 16 18 22; do
    SCD_GEMM_X=$x timeout -k 10 120 python tools/gemm_energy.py $shape 2>&1 | grep "^X=" || exit 1
  done
done; } > $O/r04_gemm_energy.txt
cat $O/r04_gemm_energy.txt
