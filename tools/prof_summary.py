#!/usr/bin/env python3
"""Compact per-kernel summary of a rocprofv3 --kernel-trace --stats csv (short names)."""
import csv, glob, re, sys
d = sys.argv[1]
f = sorted(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("ep::", "")
    return n[:60]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
for r in rows:
    if float(r["Percentage"]) < 0.05: continue
    print(f"{short(r['Name']):60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['Percentage']):6.2f}")
