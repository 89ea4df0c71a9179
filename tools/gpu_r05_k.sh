#!/bin/bash
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lockstep_merged or incremental" > $out/r05_tests_k.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 2 $out/r05_tests_k.txt
if [ $rc -ne 0 ]; then tail -n 60 $out/r05_tests_k.txt; exit 1; fi
bash tools/gpu_prof_any.sh lloyd_merged tools/lloyd_multi_prof.py 3 | cut -c1-150 | head -12
run() { # tag -- bench args
  local tag=$1; shift
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/ls.json 2> $out/ls.err || { tail -n 5 $out/ls.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/ls.json")); print("$tag:", d["value"], d["stage_ms_per_step"])
PY
}
run "c2 merged filter + refine, 4 streams"
export SCD_ESTEP_MERGED=0; run "c2 per-restart filters, 4 streams"; unset SCD_ESTEP_MERGED
run "c2 merged filter + refine, 4 streams (again)"
