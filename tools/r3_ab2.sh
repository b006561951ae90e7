#!/usr/bin/env bash
# Round-3 A/B on the GPU box.  usage (via gpurun): bash tools/r3_ab2.sh TAG "<EP_INPASS masks>" "<workloads>" [pytest targets...]
set -uo pipefail
tag="${1:-x}"; masks="${2:-0 7}"; wls="${3:-c2 ns}"; shift 3 || true
out="gpurun_out/r3_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$#" -gt 0 ]; then
  timeout 1500 python -m pytest "$@" -x -q > "$out/tests.log" 2>&1
  echo "tests rc $?"; tail -4 "$out/tests.log"
fi
quick="--no-cpu-baseline --no-bf16-secondary --no-north-star --no-configs --no-through-engine"
for wl in $wls; do
  for m in $masks; do
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
if [ -n "${EP_TRACE_MASK:-}" ]; then
  EP_INPASS=$EP_TRACE_MASK rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 30 --warmup 5 $quick --kernel-iters 5 > "$out/bench_under_trace.json" 2> "$out/trace.log"
  python3 tools/prof_summary.py "$out/trace" > "$out/kernel_stats_summary.txt"
  python3 tools/step_timeline.py "$out/trace" 60 > "$out/step_timeline.txt" 2>/dev/null
  cat "$out/step_timeline.txt"
  find "$out/trace" -name "*kernel_trace.csv" -size +20M -delete
fi
