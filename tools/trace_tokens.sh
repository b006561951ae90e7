#!/usr/bin/env bash
# rocprofv3 kernel trace of bench.py for one workload / token storage type -> kernel stats summary + one step's timeline.
# usage (via gpurun): bash tools/trace_tokens.sh TAG WORKLOAD TOKENS ["<env>"] [extra bench args]
set -uo pipefail
tag="${1:-tr}"; wl="${2:-c2}"; tok="${3:-bf16}"; envs="${4:-EP_X=1}"
if [ $# -ge 4 ]; then shift 4; else shift $#; fi
out="gpurun_out/tr_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
quick="--no-cpu-baseline --no-bf16-secondary --no-north-star --no-configs --no-through-engine"
export $envs
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 30 --warmup 5 $quick --kernel-iters 5 --workload $wl --tokens $tok "$@" > "$out/bench_under_trace.json" 2> "$out/trace.log"
python3 tools/prof_summary.py "$out/trace" > "$out/kernel_stats_summary.txt"
python3 tools/step_timeline.py "$out/trace" 60 > "$out/step_timeline.txt" 2>/dev/null
echo "== $tag ($wl, $tok, $envs)"; cat "$out/step_timeline.txt"
find "$out/trace" -name "*kernel_trace.csv" -size +20M -delete
