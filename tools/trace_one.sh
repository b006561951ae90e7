#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace of a short bench.py run -> one steady-state step timeline.
# usage: [ENV=...] bash tools/trace_one.sh TAG [bench args...]   -> gpurun_out/trace_TAG/{step_timeline.txt,kernel_stats_summary.txt}
set -uo pipefail
tag="$1"; shift
out="gpurun_out/trace_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="${EP_PROF_STEPS:---steps 30 --warmup 5} --no-cpu-baseline --no-north-star --no-configs --no-through-engine --no-bf16-secondary --kernel-iters 3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/bench_under_trace.json" 2> "$out/trace.log"
python3 tools/prof_summary.py "$out/trace" > "$out/kernel_stats_summary.txt"
python3 tools/step_timeline.py "$out/trace" ${EP_PROF_TIMELINE_STEP:-60} > "$out/step_timeline.txt" 2>/dev/null
rm -rf "$out/trace"
cat "$out/step_timeline.txt"
