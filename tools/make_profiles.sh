#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM traffic counters of bench.py.
# usage: tools/make_profiles.sh TAG [bench args...]   -> gpurun_out/profiles_TAG/*
set -uo pipefail
tag="$1"; shift
out="gpurun_out/profiles_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="${EP_PROF_STEPS:---steps 30 --warmup 5} --no-cpu-baseline --no-north-star --no-configs --no-through-engine --no-bf16-secondary --kernel-iters 10 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/bench_under_trace.json" 2> "$out/trace.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $args > /dev/null 2> "$out/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py $args > /dev/null 2> "$out/pmc_write.log"
python3 tools/prof_summary.py "$out/trace" > "$out/kernel_stats_summary.txt"
python3 tools/step_timeline.py "$out/trace" ${EP_PROF_TIMELINE_STEP:-60} > "$out/step_timeline.txt" 2>/dev/null
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
sys.path.insert(0, "tools")
from src_hash import source_hash
short = lambda n: n.split("(")[0].replace("void ", "").replace("ep::", "")
# A token pass launched INSIDE a train step follows a kernel of the step (the second pass: BatchNorm backward / dP / delta; the
# first: the previous step's update or a plane split); the stand-alone launches of the `roofline.alone` probe and of the eval
# forward follow another pool / reduce kernel.  `in_step` averages only the former (the 1.23 x mixed both).
STEP_PRED = ("bn_bwd", "gemm", "delta", "dp_thin", "opt_update", "planes_split")
for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    agg, agg_in = collections.defaultdict(list), collections.defaultdict(list)
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        prev = ""
        for r in rows:
            k = short(r["Kernel_Name"])
            agg[k].append(float(r["Counter_Value"]))
            if "pool_" in k and any(t in prev for t in STEP_PRED):
                agg_in[k].append(float(r["Counter_Value"]))
            prev = k
    res[name] = {k: {"launches": len(v), "avg_KiB": sum(v) / len(v),
                     **({"in_step_launches": len(agg_in[k]), "in_step_avg_KiB": sum(agg_in[k]) / len(agg_in[k])} if agg_in.get(k) else {})}
                 for k, v in agg.items() if "ep_" in k}
# gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> x2 (MI355X_MICROARCH.md, HBM)
summary = {}
for k, v in res["FETCH_SIZE"].items():
    w = res["WRITE_SIZE"].get(k, {"avg_KiB": 0.0})
    summary[k] = {"fetch_bytes_corrected": v["avg_KiB"] * 1024 * 2, "write_bytes": w["avg_KiB"] * 1024,
                  "hbm_bytes_per_launch": v["avg_KiB"] * 1024 * 2 + w["avg_KiB"] * 1024, "launches": v["launches"]}
    if "in_step_avg_KiB" in v:
        summary[k]["in_step"] = {"launches": v["in_step_launches"],
                                 "hbm_bytes_per_launch": v["in_step_avg_KiB"] * 1024 * 2 + w.get("in_step_avg_KiB", w["avg_KiB"]) * 1024}
json.dump({"source_hash": source_hash(), "raw": res, "per_kernel": summary,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); WRITE_SIZE as is"},
          open(f"{out}/hbm_traffic.json", "w"), indent=1)
for k, v in summary.items():
    if "pool" in k: print(k[:50], {a: round(b) for a, b in v.items()})
PY
rm -rf "$out"/pmc_fetch/*/*agent_info.csv "$out"/pmc_write/*/*agent_info.csv
cat "$out/kernel_stats_summary.txt" | head -30
