#!/bin/bash
# round-3 evidence, part A: full GPU suite, default bench line, rocprofv3 kernel stats of the same command
set -u
R=$PWD; out=$R/gpurun_out/r03; mkdir -p $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/r03_gputest.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r03_gputest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 500 python bench.py > $out/r03_bench_full.json 2> $out/bench.err; echo "[bench] rc=$?"
python - <<PY
import json
d=json.load(open("$out/r03_bench_full.json"))
print(d["value"], d["stage_ms_per_step"], d["roofline"]["frac"])
for s in d["secondary_rooflines"]: print("  ", {k:v for k,v in s.items() if k!="kernel"}, s["kernel"][:40])
print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds_per_leg_incl_sweep"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof_bench --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/prof_bench.log 2>&1
echo "[rocprof bench] rc=$?"
f=$(find $out/prof_bench -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $out/r03_bench_default_kernel_stats.csv && head -n 8 $out/r03_bench_default_kernel_stats.csv | cut -c1-160
rm -rf $out/prof_bench
