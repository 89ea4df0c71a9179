#!/bin/bash
# round-5 fifth call: c3 vote-loop sizes; the two-chain E-step against the one-chain (same box); A/B of the epilogue timing probes
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
timeout -k 10 300 python tools/c3_debug.py 2>&1 | grep -v Warning | tail -n 40
for lib in default scd_amd/lib/libscd_hip_chains2.so default scd_amd/lib/libscd_hip_chains2.so; do
  if [ $lib = default ]; then unset SCD_HIP_LIB; else export SCD_HIP_LIB=$PWD/$lib; fi
  echo "== E-step, lib $lib"
  timeout -k 10 200 python tools/estep_scaling.py 512 100 2>&1 | tail -n 11
  timeout -k 10 200 python tools/estep_scaling.py 768 100 2>&1 | tail -n 4
done > $out/r05_estep_chains_ab.txt 2>&1
unset SCD_HIP_LIB
cat $out/r05_estep_chains_ab.txt
bash tools/gpu_r04_abn.sh scd_amd/lib/libscd_hip_ablstats.so scd_amd/lib/libscd_hip_ablpre.so 2>&1 | tee $out/r05_epilogue_probes_ab.txt
