#!/bin/bash
# round-5 final evidence, part b (the tree after the last changes): full GPU suite; the two-rank gloo rehearsal of the whole job; the C4-shaped
# 1-GPU line; the --cluster KM line
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/r05_gputest.txt 2>&1; rc=$?
echo "[pytest] rc=$rc"; tail -n 3 $out/r05_gputest.txt
if [ $rc -ne 0 ]; then tail -n 60 $out/r05_gputest.txt; exit 1; fi
fi
SCD_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > $out/rehearsal.out 2> $out/rehearsal.err; echo "[rehearsal] rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$out/rehearsal.out") if l.startswith("{")][-1])
json.dump(d, open("$out/r05_bench_gpus2_gloo_rehearsal.json","w"))
print(d["n_gpus"], d["value"], d["stage_ms_per_step"], d["synthetic_name_accuracy"])
PY
timeout -k 10 600 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $out/r05_bench_c4_1gpu.json 2> $out/c4.err; echo "[c4] rc=$?"
python - <<PY
import json
d=json.load(open("$out/r05_bench_c4_1gpu.json")); print(d["value"], d["stage_ms_per_step"])
for s in d["secondary_rooflines"]: print("  ", {k:v for k,v in s.items() if k not in ("kernel","note")}, s["kernel"][:40])
PY
timeout -k 10 600 python bench.py --cluster KM --steps 2 --warmup 1 --no-cpu-baseline > $out/r05_bench_km.json 2> $out/km.err; echo "[km] rc=$?"
python - <<PY
import json
d=json.load(open("$out/r05_bench_km.json")); print(d["value"], d["stage_ms_per_step"])
for s in d["secondary_rooflines"][-2:]: print("  ", {k:v for k,v in s.items() if k not in ("kernel","note")}, s["kernel"][:40])
PY
timeout -k 10 600 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/r05_bench_c3.json 2> $out/bench_c3.err; echo "[c3] rc=$?"
python - <<PY
import json
d=json.load(open("$out/r05_bench_c3.json")); print(d["value"], d["stage_ms_per_step"], d.get("consskm"))
PY
