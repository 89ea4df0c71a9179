"""Average the counter_collection.csv files of rocprofv3 --pmc runs per kernel: python tools/pmc_summary.py DIR [substr]."""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if sub in kn:
            acc[kn[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kn, cs in acc.items():
    print(kn)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
