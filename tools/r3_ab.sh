#!/usr/bin/env bash
# Round-3 A/B on the GPU box: in-pass contractions on / off (EP_INPASS mask), tests first.
# usage (via gpurun): bash tools/r3_ab.sh TAG "<mask list>" [pytest targets...]
set -uo pipefail
tag="${1:-a}"; masks="${2:-0 3 1 2 0 3}"; shift 2 || true
out="gpurun_out/r3_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$#" -gt 0 ]; then
  timeout 1500 python -m pytest "$@" -x -q > "$out/tests.log" 2>&1
  echo "tests rc $?"; tail -5 "$out/tests.log"
fi
common="--steps 100 --warmup 10 --no-cpu-baseline --no-bf16-secondary --no-north-star --kernel-iters 5"
for wl in c2 ns; do
  i=0
  for m in $masks; do
    i=$((i+1))
    EP_INPASS=$m timeout 300 python bench.py $common --workload $wl > "$out/bench_${wl}_ip${m}_$i.json" 2> "$out/bench_${wl}_ip${m}_$i.err"
    python3 - "$out/bench_${wl}_ip${m}_$i.json" "$wl ip=$m" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "img/s", round(d["value"]), "p50", d.get("step_ms_p50"), "loss", d["check"]["mean_loss_over_timed_steps"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
# timeline of the default configuration
for wl in c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$wl" -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-north-star --no-bf16-secondary --kernel-iters 5 --workload $wl > "$out/bench_under_trace_$wl.json" 2> "$out/trace_$wl.log"
  python3 tools/prof_summary.py "$out/trace_$wl" > "$out/${wl}_kernel_stats_summary.txt"
  python3 tools/step_timeline.py "$out/trace_$wl" 60 > "$out/${wl}_step_timeline.txt" 2>/dev/null
  cat "$out/${wl}_step_timeline.txt"
  find "$out/trace_$wl" -name "*kernel_trace.csv" -size +20M -delete
done
