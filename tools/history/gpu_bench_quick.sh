#!/bin/bash
# a short bench.py run (2 steps) with its secondary lines printed: tools/gpu_bench_quick.sh [bench args]
set -u
mkdir -p gpurun_out/r03
timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/r03/bench_quick.json 2> gpurun_out/r03/bench_quick.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_quick.json"))
print(d["value"], d["stage_ms_per_step"])
for s in d["secondary_rooflines"]:
    print("  ", {k: v for k, v in s.items() if k not in ("kernel", "note")}, s["kernel"][:60])
PY
tail -n 3 gpurun_out/r03/bench_quick.err
