#!/bin/bash
# A/B of the similarity kernels on the GPU box: parity tests first, then tools/sim_bench.py; stops after a timeout.
set -u
out=gpurun_out/sim_ab; mkdir -p $out
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -E "sim_topk\[|passed|failed|Error|error" $out/$name.log | tail -n 8
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
}
SCD_SIM_RB=${RB:-8} run test_rb8 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sim_topk or vote_loop or match_missing or textual or zero_shot"
for k in ${KS:-3 5 1}; do
  [ -n "${AB4:-}" ] && SCD_SIM_RB=1 run bench_rb4_k$k 120 python tools/sim_bench.py 126976 $k
  SCD_SIM_RB=${RB:-8} run bench_rb8_k$k 120 python tools/sim_bench.py 126976 $k
done
for x in ${XS:-}; do
  SCD_SIM_RB=8 SCD_SIM_X=$x run bench_rb8_x$x 120 python tools/sim_bench.py 126976 3
done
