"""Aggregates rocprofv3 counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt and filt not in k:
            continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    short = k[:70]
    print(f"== {short}  (dispatches per counter: {len(next(iter(d.values())))})")
    for c in sorted(d):
        v = d[c]
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}   sum {sum(v):18.1f}")
