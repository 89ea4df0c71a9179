#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sim_topk" > $O/sim_tests.txt 2>&1; rc=$?
tail -n 4 $O/sim_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/r04_bench_c4_1gpu.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
python - <<PY
import json
d=json.load(open("$O/r04_bench_c4_1gpu.json"))
print(d["value"], d["stage_ms_per_step"])
for s in d["secondary_rooflines"][:1]: print("  ", {k:v for k,v in s.items() if k not in ("kernel","note")}, s["kernel"][:50])
PY
