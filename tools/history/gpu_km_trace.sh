#!/bin/bash
# kernel timeline of the Lloyd loop (rocprofv3 kernel trace, program directly after `--`): tools/gpu_km_trace.sh [n d k]
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/km_trace; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/t --output-format csv -- python3 $R/tools/lloyd_trace.py ${1:-95000} ${2:-768} ${3:-100} > $out/run.log 2>&1
echo "[trace] rc=$?"; tail -n 3 $out/run.log
f=$(find $out/t -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python3 $R/tools/trace_gaps.py $f ${4:-70} > $out/timeline.txt && cat $out/timeline.txt
find $out/t -name "*.db" -delete; find $out/t -name "*kernel_trace.csv" -delete; find $out/t -name "*agent_info*" -delete
exit 0
