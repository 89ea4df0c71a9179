#!/bin/bash
# rocprofv3 kernel stats of the k-means tools (program directly after `--`)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/km_prof; mkdir -p $out
prof() { # tag, args...
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/$tag --output-format csv -- python3 "$@" > $out/$tag.log 2>&1
  echo "[$tag] rc=$?"; grep -E "^fit|^estep|lloyd iteration" $out/$tag.log | tail -n 4
  f=$(find $out/$tag -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv && python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("  %-80s calls %5s avg %9.1f us  min %9.1f  max %9.1f  %5.1f%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
  find $out/$tag -name "*.db" -delete; find $out/$tag -name "*kernel_trace.csv" -delete; find $out/$tag -name "*agent_info*" -delete
}
prof sskm_phases $R/tools/sskm_phases.py 95000 768 100
prof kmeans_bench_c2 $R/tools/kmeans_bench.py 768 0.8 95000 100
prof kmeans_bench_c4 $R/tools/kmeans_bench.py 512 0.8 160146 1000
