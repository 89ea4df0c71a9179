#!/bin/bash
set -u
mkdir -p gpurun_out/r05
bash tools/gpu_r04_abn.sh scd_amd/lib/libscd_hip_antln1.so 2>&1 | tee gpurun_out/r05/r05_a_nt_ln1_ab.txt
