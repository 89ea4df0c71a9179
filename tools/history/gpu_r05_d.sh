#!/bin/bash
# round-5 fourth call: similarity tests (vocabulary norm once, one init launch), bench --config c3, the lock-step Lloyd loops at the
# default and the C4 size (1 / 4 / 10 streams, and one restart after the other), PMC of the streaming E-step
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sim_topk or vote or match_missing or zero_shot or textual" > $out/r05_tests_d.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r05_tests_d.txt
if [ $rc -ne 0 ]; then tail -n 60 $out/r05_tests_d.txt; exit 1; fi
timeout -k 10 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/r05_bench_c3.json 2> $out/bench_c3.err; rc=$?
echo "[bench c3] rc=$rc"; tail -n 5 $out/bench_c3.err
python - <<PY
import json
try:
    d=json.load(open("$out/r05_bench_c3.json"))
    print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d.get("consskm"), d["vote_iters"], d["synthetic_name_accuracy"])
except Exception as e: print("no c3 line", e)
PY
[ $rc -eq 124 ] && exit 1
run() { # tag env... -- bench args
  local tag=$1; shift
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/ls.json 2> $out/ls.err || { tail -n 5 $out/ls.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/ls.json")); print("$tag:", d["value"], d["stage_ms_per_step"], [ (s.get("call_us"), s.get("frac")) for s in d["secondary_rooflines"][:1]])
PY
}
for s in 1 4 10; do export SCD_LLOYD_STREAMS=$s; run "c2 lock-step streams $s"; done
unset SCD_LLOYD_STREAMS; export SCD_LLOYD_LOCKSTEP=0; run "c2 sequential restarts"; unset SCD_LLOYD_LOCKSTEP
for s in 1 4; do export SCD_LLOYD_STREAMS=$s; run "c4 lock-step streams $s" --config c4; done
unset SCD_LLOYD_STREAMS; export SCD_LLOYD_LOCKSTEP=0; run "c4 sequential restarts" --config c4; unset SCD_LLOYD_LOCKSTEP
cd /tmp && export TMPDIR=/tmp
CTR="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR -d $out/pm_estep --output-format csv -- python3 $R/tools/estep_pmc_run.py > $out/pm_estep.log 2>&1
echo "[pmc estep] rc=$?"; tail -n 3 $out/pm_estep.log
{ echo "# rocprofv3 --kernel-trace --pmc $CTR -- python3 tools/estep_pmc_run.py   (K = 100; grids: n = 98,304 / 393,216 / 524,288 rows at D = 512, then D = 768)"; python3 $R/tools/pmc_sq_summary.py $out/pm_estep estep_stream; } > $out/r05_pmc_estep.txt
cat $out/r05_pmc_estep.txt | cut -c1-170
rm -rf $out/pm_estep
