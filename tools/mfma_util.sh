#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): matrix-core busy cycles per kernel from the SQ / GRBM counters of bench.py.
# usage: tools/mfma_util.sh TAG [bench args...]   -> gpurun_out/mfma_TAG/mfma_util.json
# utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is
# summed over the 8 XCDs, MI355X_MICROARCH.md); separate --pmc pass, no tracing domains beside it.
set -uo pipefail
tag="$1"; shift
out="gpurun_out/mfma_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="${EP_PROF_STEPS:---steps 20 --warmup 5} --no-cpu-baseline --no-bf16-secondary --no-north-star --kernel-iters 5 $*"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc" -- python3 bench.py $args > /dev/null 2> "$out/pmc.log"
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
    if "ep_" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    cyc = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8.0
    if busy <= 0:
        continue
    res[k] = {"launches": len(c["GRBM_GUI_ACTIVE"]), "mfma_busy_cycles_sum_over_simds": round(busy),
              "kernel_cycles": round(cyc), "mfma_util": round(busy / (1024.0 * cyc), 4)}
json.dump({"per_kernel": res, "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); "
           "kernels that issue no MFMA are omitted"}, open(f"{out}/mfma_util.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_util"]):
    print(f"{k[:60]:60s} {v['mfma_util']:.3f}  ({v['launches']} launches, {v['kernel_cycles']} cycles)")
PY
rm -rf "$out"/pmc/*/*agent_info.csv
