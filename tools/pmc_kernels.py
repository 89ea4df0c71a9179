"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KB per dispatch).
usage: python tools/pmc_kernels.py FETCH_DIR WRITE_DIR [name filter ...]
FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B: MI355X guide; checked in round 2 on a kernel of known traffic,
profiles/r02_pmc_fc1.json)."""
import csv, glob, sys

def load(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
filt = sys.argv[3:]
print("%-70s %6s %12s %12s %12s" % ("kernel", "calls", "fetch MB", "write MB", "traffic MB"))
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    if filt and not any(f in k for f in filt):
        continue
    f = fetch[k]
    w = write.get(k, [0.0])
    fm = 2 * sum(f) / len(f) * 1024 / 1e6
    wm = sum(w) / len(w) * 1024 / 1e6
    print("%-70s %6d %12.2f %12.2f %12.2f" % (k[:70], len(f), fm, wm, fm + wm))
