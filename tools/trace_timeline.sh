#!/usr/bin/env bash
# rocprofv3 kernel trace of bench.py under one set of environment switches -> kernel stats summary + one step's timeline.
# usage (via gpurun): bash tools/trace_timeline.sh TAG "<env>" [workload]
set -uo pipefail
tag="${1:-tr}"; envs="${2:-EP_X=1}"; wl="${3:-c2}"; out="gpurun_out/ab_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
quick="--no-cpu-baseline --no-bf16-secondary --no-north-star --no-configs --no-through-engine"
export $envs
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 30 --warmup 5 $quick --kernel-iters 5 --workload $wl ${EP_BENCH_EXTRA:-} > "$out/bench_under_trace.json" 2> "$out/trace.log"
python3 tools/prof_summary.py "$out/trace" > "$out/kernel_stats_summary.txt"
python3 tools/step_timeline.py "$out/trace" 60 > "$out/step_timeline.txt" 2>/dev/null
cat "$out/step_timeline.txt"
find "$out/trace" -name "*kernel_trace.csv" -size +20M -delete
