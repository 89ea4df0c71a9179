#!/bin/bash
# the restart's Lloyd loop in C (scd_kmeans_lloyd_run) against the Python-driven loop: parity tests, then phases with both
set -u
out=gpurun_out/km; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${KTESTS:-incremental or sskm or c4_shape or lloyd or sklearn or main_}" > $out/test_run.log 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 8 $out/test_run.log
if [ $rc -ne 0 ]; then exit 1; fi
for m in 0 1; do
  SCD_LLOYD_RUN=$m timeout -k 10 200 python tools/sskm_phases.py 95000 768 100 > $out/sskm_phases_run$m.txt 2>&1; echo "[SCD_LLOYD_RUN=$m]"; tail -n 3 $out/sskm_phases_run$m.txt | cut -c1-200
done
SCD_LLOYD_RUN=1 timeout -k 10 200 python tools/sskm_phases.py 160146 512 1000 > $out/sskm_phases_c4_run1.txt 2>&1; echo "[c4 RUN=1]"; tail -n 2 $out/sskm_phases_c4_run1.txt | cut -c1-200
