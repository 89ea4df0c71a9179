#!/bin/bash
# round-5 final evidence on the final tree: full GPU suite; the default bench line (with cpu_baseline); kernel statistics of the TIMED
# region of the bench command (marker kernels, tools/trace_window_stats.py); PMC traffic of the dominant kernel (FETCH_SIZE / WRITE_SIZE in
# separate passes); matrix-pipe utilisation of the encoder kernels; kernel statistics of an SSKM fit; the power trace
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
if [ "${SKIP_TESTS:-0}" != 1 ]; then   # (the suite and the rest together no longer fit one 1,200-s call: run the suite in a call of its own)
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/r05_gputest.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r05_gputest.txt
if [ $rc -ne 0 ]; then tail -n 60 $out/r05_gputest.txt; exit 1; fi
fi
timeout -k 10 700 python bench.py > $out/r05_bench_full.json 2> $out/bench.err; echo "[bench] rc=$?"
python - <<PY
import json
d=json.load(open("$out/r05_bench_full.json"))
print(d["value"], d["stage_ms_per_step"], d["roofline"])
for s in d["secondary_rooflines"]: print("  ", {k:v for k,v in s.items() if k not in ("kernel","note")}, s["kernel"][:40])
print("  cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("seconds_per_leg_incl_sweep"), d["cpu_baseline"].get("error"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace -d $out/prof_bench --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/prof_bench.log 2>&1
rc=$?; echo "[rocprof bench] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python3 $R/tools/trace_window_stats.py $out/prof_bench $out/r05_bench_default_kernel_stats.csv | cut -c1-150 | head -n 14
rm -rf $out/prof_bench
pass() { # ctr
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $1 -d $out/pmc_fc1_$1 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pmc_fc1_$1.log 2>&1
  local rc=$?; echo "[pmc fc1 $1] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
pass FETCH_SIZE && pass WRITE_SIZE && python3 $R/tools/pmc_traffic.py $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE 786432 3072 768 $out/r05_pmc_fc1.json 3990 | tail -n 12
rm -rf $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTR -d $out/pm_enc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pm_enc.log 2>&1
echo "[pmc mfma] rc=$?"
{ echo "# rocprofv3 --kernel-trace --pmc $CTR -- python3 bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline"; python3 $R/tools/pmc_mfma_util.py $out/pm_enc gemm_w4 attention sim_topk; } > $out/r05_pmc_mfma_encoder.txt
cat $out/r05_pmc_mfma_encoder.txt | cut -c1-140
rm -rf $out/pm_enc
bash $R/tools/gpu_prof_any.sh lloyd_fit tools/lloyd_multi_prof.py 3 | cut -c1-150 | head -n 14
cp $R/gpurun_out/prof_lloyd_fit/kernel_stats.csv $out/r05_lloyd_fit_kernel_stats.csv
cd $R
bash tools/gpu_power_trace.sh > $out/power_trace.log 2>&1; tail -n 3 $out/power_trace.log
