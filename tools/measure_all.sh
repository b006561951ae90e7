#!/usr/bin/env bash
# Run ON THE GPU BOX: one bench line per BASELINE config (EP head) and per registry head, short form -> OUT (jsonl)
out="${1:-gpurun_out/all_heads.jsonl}"; mkdir -p "$(dirname "$out")"; : > "$out"
base="--no-cpu-baseline --no-north-star --no-bf16-secondary --no-configs --steps ${STEPS:-60} --warmup 10"
common="$base --no-through-engine"
for w in c1 c2 ns c3 c4 c5; do python3 bench.py $common --workload $w 2>/dev/null | tail -1 >> "$out"; done
# (the heads keep the train_one_epoch leg: the same workload read in place from a resident store, with the store's cached tables)
for h in coca siglip cae jepa aim simpool esimpool cait clip cbam dolg dinovit abmilp; do python3 bench.py $base --engine-steps 100 --head $h 2>/dev/null | tail -1 >> "$out"; done
python3 bench.py $common --head coca --workload c4 2>/dev/null | tail -1 >> "$out"
python3 bench.py $common --head abmilp --workload c4 2>/dev/null | tail -1 >> "$out"
python3 - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    t = d.get("train_one_epoch") or {}
    print("%-34s %-60s %12.0f img/s  %8.4f ms%s" % (d["metric"], d["config"]["workload"][:60], d["value"], d["ms_per_step"],
          ("   | over a resident store: %10.0f img/s  %8.4f ms" % (t["value"], t["ms_per_step"])) if t else ""))
PY
