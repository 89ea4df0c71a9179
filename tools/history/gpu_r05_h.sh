#!/bin/bash
# round-5: the merged multi-restart filter (estep_rbm_kernel) - tests, stage times; bench --config c3
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lockstep_merged or incremental or lloyd_run or sskm_matches or two_ranks or multi_rank or sharded_loops or workspaces" > $out/r05_tests_h.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r05_tests_h.txt
if [ $rc -ne 0 ]; then tail -n 60 $out/r05_tests_h.txt; exit 1; fi
run() { # tag -- bench args
  local tag=$1; shift
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/ls.json 2> $out/ls.err || { tail -n 5 $out/ls.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/ls.json")); print("$tag:", d["value"], d["stage_ms_per_step"])
PY
}
# (labels as of the tree this script was written for, where the merged filter was the default; since the end of round 5 it is opt-in:
# SCD_ESTEP_MERGED=1 selects it, unset / 0 = per-restart filters.  The runs below say what they set.)
export SCD_ESTEP_MERGED=1
run "c2 merged filter (SCD_ESTEP_MERGED=1), 4 streams"
export SCD_LLOYD_STREAMS=10; run "c2 merged filter (SCD_ESTEP_MERGED=1), 10 streams"; unset SCD_LLOYD_STREAMS
export SCD_LLOYD_STREAMS=1; run "c2 merged filter (SCD_ESTEP_MERGED=1), 1 stream"; unset SCD_LLOYD_STREAMS
unset SCD_ESTEP_MERGED; run "c2 per-restart filters (default), 4 streams"
timeout -k 10 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/r05_bench_c3.json 2> $out/bench_c3.err; rc=$?
echo "[bench c3] rc=$rc"; tail -n 5 $out/bench_c3.err
python - <<PY
import json
try:
    d=json.load(open("$out/r05_bench_c3.json"))
    print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d.get("consskm"), d["vote_iters"], d["synthetic_name_accuracy"])
except Exception as e: print("no c3 line", e)
PY
