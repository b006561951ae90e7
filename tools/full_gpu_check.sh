#!/usr/bin/env bash
# Full GPU test suite + the default bench line, as the driver runs them.  usage (via gpurun): bash tools/full_gpu_check.sh TAG
set -uo pipefail
tag="${1:-full}"; out="gpurun_out/ab_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -q -m gpu -x ) > "$out/gpu_tests.log" 2>&1; echo "gpu tests rc $?"; tail -5 "$out/gpu_tests.log"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc $?"
tail -3 "$out/bench_default.err"; tail -c 6000 "$out/bench_default.json"
