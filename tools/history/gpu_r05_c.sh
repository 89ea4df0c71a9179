#!/bin/bash
# round-5 third call: LayerNorm row pairs precomputed (ln_finish_kernel) - encoder tests, same-box A/B against the previous encoder build;
# the restarts' Lloyd loops in lock-step (scd_kmeans_lloyd_run_multi) - k-means tests, stage time with 1 / 4 / 10 streams; DINO features of
# the synthetic images (bench --config c3)
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sskm or lloyd or incremental or two_ranks or sharded_loops or multi_rank or c1_shape or workspaces or main_unsup or kpp" > $out/r05_tests_c.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r05_tests_c.txt
if [ $rc -ne 0 ]; then tail -n 60 $out/r05_tests_c.txt; exit 1; fi
bash tools/gpu_r04_abn.sh scd_amd/lib/libscd_hip_r04enc.so 2>&1 | tee $out/r05_ln_finish_ab.txt
for s in 1 4 10; do
  SCD_LLOYD_STREAMS=$s timeout -k 10 300 python bench.py --steps 3 --warmup 1 --images 15960 --no-cpu-baseline > $out/ls.json 2> $out/ls.err || { tail -n 5 $out/ls.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/ls.json")); print("lock-step streams $s:", d["stage_ms_per_step"])
PY
done
SCD_LLOYD_LOCKSTEP=0 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --images 15960 --no-cpu-baseline > $out/ls.json 2> $out/ls.err || exit 1
python - <<PY
import json
d=json.load(open("$out/ls.json")); print("sequential restarts:", d["stage_ms_per_step"])
PY
timeout -k 10 300 python tools/c3_feature_check.py 2>&1 | tail -n 5
