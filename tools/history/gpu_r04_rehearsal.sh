#!/bin/bash
# two ranks sharing the one GPU, gloo collectives: the N = 2 job end to end (multi-GPU code path rehearsal)
set -u
O=gpurun_out/r04; mkdir -p $O
SCD_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_gpus2_gloo_rehearsal.json 2> $O/rehearsal.err; rc=$?
echo "rc=$rc"; tail -n 5 $O/rehearsal.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_gpus2_gloo_rehearsal.json") if l.startswith("{")][-1])   # gloo prints its own lines to stdout
print(d["n_gpus"], d["value"], d["stage_ms_per_step"], d["synthetic_name_accuracy"])
PY
