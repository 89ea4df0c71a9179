"""Timeline of the last kernels of a rocprofv3 --kernel-trace csv: start (us, relative), duration, gap to the previous kernel's end.
python tools/trace_gaps.py kernel_trace.csv [count] [name filter for the anchor kernel]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-cnt:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:70]))
    prev_end = e
