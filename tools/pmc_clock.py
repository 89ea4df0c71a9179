"""Effective shader clock per kernel from a `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` run:
clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md 'DVFS give-back').
  python tools/pmc_clock.py DIR [out.json]
Dispatches are listed in launch order so that they can be matched with the operand fill the launching script used."""
import csv
import glob
import json
import sys

d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        ns = None
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if (not ns or ns <= 0) and r["Dispatch_Id"] in dur:
            ns = dur[r["Dispatch_Id"]][0]
        if not ns or ns < 200000:            # the quotient reads high on dispatches shorter than ~0.3 ms
            continue
        rows.append(dict(dispatch=int(r["Dispatch_Id"]), kernel=r["Kernel_Name"][:70], us=round(ns / 1e3, 1),
                         ghz=round(float(r["Counter_Value"]) / 8.0 / ns, 3)))
rows.sort(key=lambda x: x["dispatch"])
for r in rows:
    print("%6d %-70s %10.1f us  %.3f GHz" % (r["dispatch"], r["kernel"], r["us"], r["ghz"]))
if len(sys.argv) > 2:
    json.dump(rows, open(sys.argv[2], "w"), indent=1)
