#!/usr/bin/env python3
"""Average PMC counters per kernel from rocprofv3 counter_collection csv files: pmc_summary.py DIR [substr]"""
import csv, glob, collections, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "pool"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ep::", "")[:48]
        if sub in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
