#!/bin/bash
# round-4 final evidence: full GPU suite, the default bench line (with cpu_baseline), PMC traffic of the dominant kernel
set -u
R=$PWD; out=$R/gpurun_out/r04; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/r04_gputest.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r04_gputest.txt
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 700 python bench.py > $out/r04_bench_full.json 2> $out/bench.err; echo "[bench] rc=$?"
python - <<PY
import json
d=json.load(open("$out/r04_bench_full.json"))
print(d["value"], d["stage_ms_per_step"], d["roofline"])
for s in d["secondary_rooflines"]: print("  ", {k:v for k,v in s.items() if k not in ("kernel","note")}, s["kernel"][:40])
print("  cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("seconds_per_leg_incl_sweep"), d["cpu_baseline"].get("error"))
PY
cd /tmp && export TMPDIR=/tmp
pass() { # ctr
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $1 -d $out/pmc_fc1_$1 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pmc_fc1_$1.log 2>&1
  local rc=$?; echo "[pmc fc1 $1] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
pass FETCH_SIZE && pass WRITE_SIZE && python3 $R/tools/pmc_traffic.py $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE 786432 3072 768 $out/r04_pmc_fc1.json 3990 | tail -n 30
rm -rf $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE
cd $R
bash tools/gpu_power_trace.sh > $out/power_trace.log 2>&1; tail -n 3 $out/power_trace.log
