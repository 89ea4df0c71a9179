#!/bin/bash
# round 6: the stress / fuzz runs after the E-step prep, distance kernel and seeding changes
set -u
out=gpurun_out/r06; mkdir -p $out
timeout -k 10 500 python tests/stress_estep.py 100 > $out/r06_stress_estep.txt 2>&1; echo "[stress_estep] rc=$?"; tail -n 6 $out/r06_stress_estep.txt
timeout -k 10 400 python tests/stress_lockstep.py > $out/r06_stress_lockstep.txt 2>&1; echo "[stress_lockstep] rc=$?"; tail -n 4 $out/r06_stress_lockstep.txt
timeout -k 10 600 python tools/hip_fuzz_kmeans.py 900 14 > $out/r06_hip_fuzz_kmeans.txt 2>&1; echo "[fuzz] rc=$?"; tail -n 4 $out/r06_hip_fuzz_kmeans.txt
