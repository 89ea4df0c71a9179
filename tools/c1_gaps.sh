#!/bin/bash
# kernel timeline of the greedy k-means++ rounds inside `bench.py --config c1`: durations and gaps between dependent launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r06/c1_trace; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace -d $o --output-format csv -- python3 $R/bench.py --config c1 --steps 1 --warmup 1 --no-cpu-baseline > $o/run.log 2>&1 || { tail $o/run.log; exit 1; }
f=$(find $o -name "*kernel_trace.csv" | head -n 1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: between the marker kernels
b = max(i for i, r in enumerate(rows) if "scd_mark_begin" in r["Kernel_Name"]); e = max(i for i, r in enumerate(rows) if "scd_mark_end" in r["Kernel_Name"])
rows = rows[b:e + 1]
ks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("kg_select")]
seg = rows[ks[50]:ks[53] + 1]          # three consecutive rounds in the middle of the seeding
prev = None
for r in seg:
    s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("dur %7.1f  gap %6.1f  %s" % ((en - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r["Kernel_Name"][:60]))
    prev = en
tot_k = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[ks[0]:ks[-1]]) / 1e3
span = (int(rows[ks[-1]]["End_Timestamp"]) - int(rows[ks[0]]["Start_Timestamp"])) / 1e3
print("seeding rounds %d: span %.1f us, kernel time %.1f us, launches %d" % (len(ks) - 1, span, tot_k, ks[-1] - ks[0]))
PY
rm -rf $o
