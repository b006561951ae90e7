#!/usr/bin/env bash
# Run ON THE GPU BOX: one PMC pass per counter group over a command; prints per-kernel averages.
# usage: tools/exp_pmc.sh OUTDIR "CTR1 CTR2" -- python3 script args...
out="$1"; ctrs="$2"; shift 3; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc" -- "$@" > /dev/null 2> "$out/pmc.log"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ep::", "")[:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "ep_" not in k: continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", max(len(v) for v in d.values()))
PY
rm -rf "$out/pmc"
