#!/bin/bash
# round 4, first GPU session: write-rate micro, GEMM stagger experiment (ablate build), quick bench baseline
set -u
O=gpurun_out/r04
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/hbm_write tools/micro/hbm_write_rate.hip && timeout -k 10 120 /tmp/hbm_write > $O/hbm_write_rate.txt 2>&1
echo "write micro rc=$?"
for st in 0 100 300 1048676 1048876 1049176; do
  echo "== SCD_GEMM_STAGGER=$st" >> $O/gemm_stagger.txt
  SCD_HIP_LIB=scd_amd/lib/libscd_hip_ablate.so SCD_GEMM_STAGGER=$st timeout -k 10 200 python tools/gemm_bench.py 665 >> $O/gemm_stagger.txt 2>&1 || exit 1
done
echo "stagger done"
bash tools/gpu_bench_quick.sh && cp gpurun_out/r03/bench_quick.json $O/bench_quick_baseline.json
