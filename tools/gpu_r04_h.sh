#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip_towers or encoder or dino or extract_feature or text_tower or estep or sskm or lloyd or incremental" > $O/enc_tests.txt 2>&1; rc=$?
tail -n 3 $O/enc_tests.txt
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_r04_ab.sh scd_amd/lib/libscd_hip_prev.so
for lib in default scd_amd/lib/libscd_hip_prev.so; do
  if [ $lib = default ]; then unset SCD_HIP_LIB; else export SCD_HIP_LIB=$PWD/$lib; fi
  echo "== $lib"; timeout -k 10 200 python tools/kmeans_bench.py 768 0.8 95000 100 2>&1 | grep -e "^estep" | head -3
done
