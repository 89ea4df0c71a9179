"""Per-kernel statistics of the TIMED region of a bench.py run, from a rocprofv3 --kernel-trace output directory: only the dispatches between
scd_mark_begin_kernel and scd_mark_end_kernel (bench.py launches them around its timed steps) are counted, so the set-up work of the
command (synthetic images, vocabulary, warm-up steps: ATen kernels among them) cannot be confused with the measured path.
    python tools/trace_window_stats.py DIR [out.csv]"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
b = [int(r["Start_Timestamp"]) for r in rows if "scd_mark_begin_kernel" in r["Kernel_Name"]]
e = [int(r["Start_Timestamp"]) for r in rows if "scd_mark_end_kernel" in r["Kernel_Name"]]
if not b or not e:
    sys.exit("no marker kernels in the trace")
t0, t1 = b[-1], e[-1]
acc = collections.defaultdict(list)
for r in rows:
    s = int(r["Start_Timestamp"])
    if t0 < s < t1 and "scd_mark_" not in r["Kernel_Name"]:
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - s)
tot = sum(sum(v) for v in acc.values())
lines = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    lines.append((k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 3), min(v), max(v)))
print("# timed region: %.3f ms between the markers, %.3f ms of kernel time in %d launches of %d kernels" % ((t1 - t0) / 1e6, tot / 1e6, sum(len(v) for v in acc.values()), len(acc)))
for ln in lines[:24]:
    print("%-96s %7s %14s %12s %8s" % (str(ln[0])[:96], ln[1], ln[2], ln[3], ln[4]))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as f:
        f.write("# rocprofv3 --kernel-trace of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`, dispatches between scd_mark_begin_kernel and scd_mark_end_kernel only (tools/trace_window_stats.py)\n")
        csv.writer(f).writerows(lines)
