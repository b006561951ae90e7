#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): where the waves of each kernel spend their cycles (SQ counters, one --pmc pass).
# usage: tools/wave_states.sh TAG [bench args...]  -> gpurun_out/waves_TAG/wave_states.json
# SQ_WAIT_ANY = parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three add up
# to about SQ_WAVE_CYCLES (MI355X_MICROARCH.md, PMC table).
set -uo pipefail
tag="$1"; shift
out="gpurun_out/waves_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="${EP_PROF_STEPS:---steps 20 --warmup 5} --no-cpu-baseline --no-bf16-secondary --no-north-star --kernel-iters 5 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$out/pmc" -- python3 bench.py $args > /dev/null 2> "$out/pmc.log"
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ep::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in agg.items():
    if "ep_pool" not in k and "ep_gemm" not in k:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    res[k] = {"launches": len(c["SQ_WAVE_CYCLES"])}
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if n in m:
            res[k][n + "/WAVE_CYCLES"] = round(m[n] / wc, 4)
json.dump({"per_kernel": res, "note": "fractions of SQ_WAVE_CYCLES, averaged over the launches of a kernel"},
          open(f"{out}/wave_states.json", "w"), indent=1)
for k, v in res.items():
    print(k[:56].ljust(56), {a.split("/")[0].replace("SQ_", ""): b for a, b in v.items() if a != "launches"})
PY
rm -rf "$out"/pmc/*/*agent_info.csv
