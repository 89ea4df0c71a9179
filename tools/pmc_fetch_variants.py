"""Per-kernel FETCH_SIZE (x2: gfx950 tallies every request pattern we use at half, profiles/r04_fetch_calib.txt) of the full-size launches of a
rocprofv3 --pmc FETCH_SIZE pass: python tools/pmc_fetch_variants.py DIR [substring ...]"""
import csv, glob, sys
d, subs = sys.argv[1], sys.argv[2:] or ["gemm_w4", "attention_persist"]
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and any(s in row["Kernel_Name"] for s in subs):
            acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    m = max(v)
    sel = [x for x in v if x > 0.9 * m]
    print("%-70s full launches %4d  fetch beyond L2 %8.1f MB per launch" % (k[:70], len(sel), 2 * 1024 * sum(sel) / len(sel) / 1e6))
