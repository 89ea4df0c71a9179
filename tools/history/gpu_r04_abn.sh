#!/bin/bash
# same-box comparison of several library builds: tools/gpu_r04_abn.sh LIB ... (each once per repetition, two repetitions, default first)
set -u
O=gpurun_out/r04; mkdir -p $O
for rep in 1 2; do
for lib in default "$@"; do
  if [ $lib = default ]; then unset SCD_HIP_LIB; else export SCD_HIP_LIB=$PWD/$lib; fi
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || exit 1
  python - <<PY
import json
d=json.load(open("$O/ab.json"))
print("rep $rep lib %-36s %9.1f images/s  encode %.1f ms  fc1 frac %.4f" % ("$lib", d["value"], d["stage_ms_per_step"]["encode"], d["roofline"]["frac"]))
PY
done
done
