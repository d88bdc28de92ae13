"""Mean per-launch PMC counter values per kernel from a rocprofv3 --pmc output directory (counter_collection.csv)."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "smr::" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k, {c: "%.4g" % (acc[k][c] / n[k][c]) for c in sorted(acc[k])}, "launches", max(n[k].values()))
