#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c3_shape or c1_shape" --durations=5 > $O/c3_test.txt 2>&1; rc=$?
tail -n 12 $O/c3_test.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/kmeans_bench.py 512 0.8 160146 1000 > $O/r04_kmeans_bench_c4.txt 2>&1; echo "kmeans_bench c4 rc=$?"; tail -n 12 $O/r04_kmeans_bench_c4.txt
timeout -k 10 600 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/r04_bench_c4_1gpu.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
python - <<PY
import json
d=json.load(open("$O/r04_bench_c4_1gpu.json"))
print(d["value"], d["stage_ms_per_step"])
for s in d["secondary_rooflines"]: print("  ", {k:v for k,v in s.items() if k not in ("kernel","note")}, s["kernel"][:50])
PY
