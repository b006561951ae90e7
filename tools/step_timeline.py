#!/usr/bin/env python3
"""Print the kernel timeline of one steady-state step from a rocprofv3 kernel_trace csv directory."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
isfwd = lambda r: "pool_" in r["Kernel_Name"] and "fwd" in r["Kernel_Name"]
idx = [i for i, r in enumerate(rows) if isfwd(r) and not (i > 0 and isfwd(rows[i - 1]))]     # (a pass in query chunks: its first launch)
if len(idx) < 3:                      # heads without a token pass (AbMILP, DINOv2 block, DOLG): a step ends with the optimizer's update
    idx = [i + 1 for i, r in enumerate(rows) if "opt_update" in r["Kernel_Name"]][:-1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else min(20, len(idx) - 6)
k = max(0, min(k, len(idx) - 2))
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ep::", "")[:36]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f %7.1f  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n))
