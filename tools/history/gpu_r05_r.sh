#!/bin/bash
set -u
mkdir -p gpurun_out/r05
{ timeout -k 10 500 python tests/stress_lockstep.py 12; echo "rc=$?"; timeout -k 10 500 python tests/stress_determinism.py 10 3990; echo "rc=$?"; } 2>&1 | grep -v Warning | tee gpurun_out/r05/r05_stress.txt | tail -n 12
