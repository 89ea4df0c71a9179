#!/bin/bash
# round-5 second call: the new tests (outlier-weight towers, configs[2] with ConSSKM end to end, the constrained engine), the
# configs[2] bench line, and a same-box A/B of the LayerNorm-fold timing probes (-DW4_LN_ABL=1/2: results wrong, timing only)
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "outlier or consskm or constrained or c3_shape" -s > $out/r05_newtests.txt 2>&1; rc=$?
echo "[pytest new] rc=$rc"; grep -E "C3 ConSSKM fit|passed|failed|Error|error" $out/r05_newtests.txt | tail -n 12
if [ $rc -ne 0 ]; then tail -n 40 $out/r05_newtests.txt; exit 1; fi
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
bash tools/gpu_r04_abn.sh scd_amd/lib/libscd_hip_lnabl1.so scd_amd/lib/libscd_hip_lnabl2.so 2>&1 | tee $out/r05_ln_ablation_ab.txt
