#!/bin/bash
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
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
bash tools/gpu_r04_abn.sh scd_amd/lib/libscd_hip_ablstats.so scd_amd/lib/libscd_hip_ablpre.so 2>&1 | tee $out/r05_epilogue_probes_ab.txt
tail -n 5 gpurun_out/r04/ab.err
