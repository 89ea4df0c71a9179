#!/bin/bash
set -u
R=$PWD; out=$R/gpurun_out/r04; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/prof_km --output-format csv -- python3 $R/tools/km_fit_bench.py > $out/prof_km.log 2>&1
echo "rc=$?"
f=$(find $out/prof_km -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $out/r04_km_fit_kernel_stats.csv
rm -rf $out/prof_km
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/r04_km_fit_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6)
for r in rows[:16]:
    print("%-60s calls %6s avg %9.1f us tot %8.2f ms %5.2f%%"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
