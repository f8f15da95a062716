"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter.
usage: python profiles/pmc_summary.py <dir> [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt in k:
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
