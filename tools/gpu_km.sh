#!/bin/bash
# k-means checks on the GPU box: parity tests of the E-step paths, C4-size timing, Lloyd phases
set -u
out=gpurun_out/km; mkdir -p $out gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "estep or c4_shape or lloyd or sskm or kmeans or kpp" > $out/test.log 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 6 $out/test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for rb in 0 1; do
  SCD_ESTEP_RB=$rb timeout -k 10 200 python tools/kmeans_bench.py 512 0.8 160146 1000 > $out/c4_rb$rb.log 2>&1; echo "[c4 SCD_ESTEP_RB=$rb] rc=$?"; grep -E "^estep|lloyd iteration" $out/c4_rb$rb.log
done
timeout -k 10 200 python tools/sskm_phases.py 95000 768 100 > $out/sskm_phases.txt 2>&1; tail -n 3 $out/sskm_phases.txt
