"""Matrix-pipe utilisation per kernel from a rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE run.
SQ_VALU_MFMA_BUSY_CYCLES: matrix-pipe busy cycles summed over the 1,024 SIMDs (checked in round 1 against FLOP / FLOP-per-MFMA x cycles-per-
MFMA); SQ_BUSY_CYCLES: summed over 32 SQ instances, so kernel cycles = SQ_BUSY_CYCLES / 32; utilisation = MFMA_BUSY / (kernel cycles x 1,024);
clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration.   python tools/pmc_mfma_util.py DIR [name filter ...]"""
import csv, glob, sys, collections
d, filt = sys.argv[1], sys.argv[2:]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt and not any(x in k for x in filt):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            acc[k]["_ns"].append(dur[r["Dispatch_Id"]])
print("%-72s %5s %10s %12s %9s %8s" % ("kernel", "calls", "avg us", "MFMA busy", "util", "GHz"))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "SQ_BUSY_CYCLES" not in c:
        continue
    # full-size launches only (within 10 % of the longest)
    b = c["SQ_BUSY_CYCLES"]; m = c["SQ_VALU_MFMA_BUSY_CYCLES"]; g = c.get("GRBM_GUI_ACTIVE", []); ns = c.get("_ns", [])
    keep = [i for i in range(len(b)) if b[i] > 0.9 * max(b)]
    cyc = sum(b[i] for i in keep) / len(keep) / 32.0
    busy = sum(m[i] for i in keep) / len(keep)
    us = sum(ns[i] for i in keep if i < len(ns)) / max(1, len([i for i in keep if i < len(ns)])) / 1e3 if ns else float("nan")
    ghz = (sum(g[i] for i in keep if i < len(g)) / max(1, len(keep)) / 8.0 / (us * 1e3)) if (g and ns) else float("nan")
    if busy <= 0:
        continue
    print("%-72s %5d %10.1f %12.4g %9.3f %8.2f" % (k[:72], len(keep), us, busy, busy / (cyc * 1024.0), ghz))
