#!/usr/bin/env bash
# Run ON THE GPU BOX: one bench line per BASELINE config (EP head) and per registry head, short form -> OUT (jsonl)
out="${1:-gpurun_out/all_heads.jsonl}"; mkdir -p "$(dirname "$out")"; : > "$out"
common="--no-cpu-baseline --no-north-star --no-bf16-secondary --no-configs --no-through-engine --steps ${STEPS:-60} --warmup 10"
for w in c1 c2 ns c3 c4 c5; do python3 bench.py $common --workload $w 2>/dev/null | tail -1 >> "$out"; done
for h in coca siglip cae jepa aim simpool esimpool cait clip cbam dolg dinovit abmilp; do python3 bench.py $common --head $h 2>/dev/null | tail -1 >> "$out"; done
python3 bench.py $common --head coca --workload c4 2>/dev/null | tail -1 >> "$out"
python3 bench.py $common --head abmilp --workload c4 2>/dev/null | tail -1 >> "$out"
python3 - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print("%-34s %-60s %12.0f img/s  %8.4f ms" % (d["metric"], d["config"]["workload"][:60], d["value"], d["ms_per_step"]))
PY
