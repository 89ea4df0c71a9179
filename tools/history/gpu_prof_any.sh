#!/bin/bash
# rocprofv3 kernel stats of one python tool: tools/gpu_prof_any.sh TAG script.py [args]
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
out=$R/gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 $R/"$@" > $out/run.log 2>&1
echo "[$tag] rc=$?"; tail -n 4 $out/run.log
f=$(find $out -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("  %-84s calls %5s avg %9.1f us  min %9.1f  max %9.1f  %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info*" -delete
exit 0
