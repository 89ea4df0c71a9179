#!/bin/bash
# k-means checks on the GPU box: parity tests of the E-step / M-step paths, Lloyd phases with and without the incremental M-step
set -u
out=gpurun_out/km; mkdir -p $out gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${KTESTS:-estep or c4_shape or lloyd or sskm or kmeans or kpp or mstep or constrained or main_}" > $out/test.log 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 12 $out/test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for dl in 0 1; do
  SCD_MSTEP_DELTA=$dl timeout -k 10 200 python tools/sskm_phases.py 95000 768 100 > $out/sskm_phases_delta$dl.txt 2>&1; echo "[SCD_MSTEP_DELTA=$dl]"; tail -n 3 $out/sskm_phases_delta$dl.txt | cut -c1-330
done
