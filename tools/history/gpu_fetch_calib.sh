#!/bin/bash
# FETCH_SIZE calibration on known byte counts (tools/micro/fetch_calib.hip) -> gpurun_out/r04/r04_fetch_calib.txt
set -u
R=$PWD; out=$R/gpurun_out/r04; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fc --output-format csv -- $R/tools/micro/fetch_calib > $out/fc.log 2>&1 || { cat $out/fc.log | tail -n 5; exit 1; }
python3 - <<PY > $out/r04_fetch_calib.txt
import csv, glob
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/micro/fetch_calib  (every kernel reads each of 1,610,612,736 bytes once; FETCH_SIZE is in KB)")
B = 2048 * 512 * 1536
for f in glob.glob("$out/fc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            v = float(row["Counter_Value"]) * 1024
            print("%-60s FETCH_SIZE %8.1f MB   / bytes read = %.3f" % (row["Kernel_Name"][:60], v / 1e6, v / B))
PY
cat $out/r04_fetch_calib.txt
rm -rf $out/fc
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RD[A-Za-z0-9_]*\|TCC_EA_RD[A-Za-z0-9_]*\|TCC_MALL[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $out/tcc_counters.txt; cat $out/tcc_counters.txt
