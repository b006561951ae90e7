#!/usr/bin/env bash
# In-pass task timelines (EP_IP_STAMP=1 prints per-workgroup stamps to stderr; the launches synchronise, so the step
# time of THAT run means nothing), then a short on/off A/B.  usage (via gpurun): bash tools/r3_stamp.sh TAG
set -uo pipefail
tag="${1:-s}"; out="gpurun_out/r3_$tag"; mkdir -p "$out"
cd "$GRAFT_REPO_ROOT"
quick="--no-cpu-baseline --no-bf16-secondary --no-north-star --no-configs --no-through-engine"
EP_IP_STAMP=1 EP_INPASS=3 timeout 300 python bench.py --steps 60 --warmup 5 --spinup 10 $quick --kernel-iters 2 > "$out/stamp.json" 2> "$out/stamp.err"
grep -A3 "EP_IP_STAMP. fwd" "$out/stamp.err" | tail -8
grep -A3 "EP_IP_STAMP. bwd" "$out/stamp.err" | tail -8
for wl in c2 ns; do
  for m in ${2:-0 3 1 2 3 0}; do
    EP_INPASS=$m timeout 300 python bench.py --steps 100 --warmup 10 $quick --kernel-iters 5 --workload $wl > "$out/b_${wl}_$m.json" 2> "$out/b_${wl}_$m.err"
    python3 - "$out/b_${wl}_$m.json" "$wl ip=$m" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"].get("in_step", {})
    print(sys.argv[2], "ms/step", d["ms_per_step"], "img/s", round(d["value"]), "p50", d.get("step_ms_p50"), "in-step fwd/bwd us", r.get("fwd_us"), r.get("bwd_us"), "loss", d["check"]["mean_loss_over_timed_steps"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
