#!/bin/bash
# per-kernel durations of tools/sim_bench.py under rocprofv3 --kernel-trace --stats: tools/gpu_sim_stats.sh [k] [tag]
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
k=${1:-3}; tag=${2:-rb8}
out=$R/gpurun_out/sim_stats_$tag; mkdir -p $out
export SCD_SIM_RB=${SCD_SIM_RB:-16}
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 $R/tools/sim_bench.py 126976 $k > $out/run.log 2>&1
rc=$?; echo "[stats k=$k] rc=$rc"; grep sim_topk $out/run.log
f=$(find $out -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(t in r["Name"] for t in ("sim_", "wmax", "fill", "Memset", "memset")):
        print("%-90s calls %4s avg %10.1f us  min %10.1f  max %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info*" -delete
exit 0
