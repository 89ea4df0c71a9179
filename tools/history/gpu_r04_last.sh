#!/bin/bash
# last evidence of round 4 on the final tree: the GPU suite, the default bench line, --cluster KM, --config c4
set -u
R=$PWD; out=$R/gpurun_out/r04; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/r04_gputest.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 2 $out/r04_gputest.txt
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 700 python bench.py > $out/r04_bench_full.json 2> $out/bench.err; echo "[bench] rc=$?"
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cluster KM > $out/r04_bench_quick_km.json 2> $out/bench_km.err; echo "[bench KM] rc=$?"
timeout -k 10 500 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $out/r04_bench_c4_1gpu.json 2> $out/bench_c4.err; echo "[bench c4] rc=$?"
python - <<PY
import json
for f in ("r04_bench_full.json", "r04_bench_quick_km.json", "r04_bench_c4_1gpu.json"):
    d = json.load(open("$out/" + f))
    print(f, d["value"], d["stage_ms_per_step"], d["roofline"]["frac"], d["board_power"]["median_w"] if d.get("board_power") else None)
PY
