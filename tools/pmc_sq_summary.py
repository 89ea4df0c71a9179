"""Where a kernel's wave time goes, from ONE rocprofv3 --kernel-trace --pmc pass with
  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
(MI355X guide, PMC slots: WAIT_ANY = wave parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; the three
are disjoint and add up to ~WAVE_CYCLES; MFMA busy is summed over the SIMDs, SQ_BUSY over 32 SQ instances).
Dispatches are grouped by (kernel, grid size) so that differently sized launches of one kernel are separate lines.
    python tools/pmc_sq_summary.py DIR [name filter ...]"""
import csv, glob, sys, collections
d, filt = sys.argv[1], sys.argv[2:]
info = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        info[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt and not any(x in k for x in filt):
            continue
        ns, grid = info.get(r["Dispatch_Id"], (0, r.get("Grid_Size", "?")))
        acc[(k, grid)][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        acc[(k, grid)][r["Dispatch_Id"]]["_ns"] = ns
print("%-46s %9s %5s %8s | %6s %6s %6s | %6s %5s | %9s %9s" % ("kernel", "grid", "calls", "avg us", "parked", "stall", "issue", "mfma", "GHz", "VALU/wave", "LDS/wave"))
for (k, grid), disp in sorted(acc.items(), key=lambda kv: -sum(v.get("_ns", 0) for v in kv[1].values())):
    rows = list(disp.values())
    n = len(rows)
    avg = lambda name: sum(r.get(name, 0.0) for r in rows) / n
    wc = avg("SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    us = avg("_ns") / 1e3
    cyc = avg("SQ_BUSY_CYCLES") / 32.0
    mf = avg("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0) if cyc > 0 else float("nan")
    ghz = avg("GRBM_GUI_ACTIVE") / 8.0 / (us * 1e3) if us > 0 else float("nan")
    waves = avg("SQ_WAVES") if any("SQ_WAVES" in r for r in rows) else 0
    print("%-46s %9s %5d %8.1f | %6.3f %6.3f %6.3f | %6.3f %5.2f | %9.4g %9.4g" % (
        k[:46], grid, n, us, avg("SQ_WAIT_ANY") / wc, avg("SQ_WAIT_INST_ANY") / wc, avg("SQ_ACTIVE_INST_ANY") / wc, mf, ghz,
        avg("SQ_INSTS_VALU") / waves if waves else avg("SQ_INSTS_VALU"), avg("SQ_INSTS_LDS") / waves if waves else avg("SQ_INSTS_LDS")))
