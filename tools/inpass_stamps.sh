#!/usr/bin/env bash
# Per-workgroup timelines of the in-pass contractions (EP_IP_STAMP=1: the launches synchronise, so the step time of these
# runs means nothing).
# usage (via gpurun): bash tools/inpass_stamps.sh TAG "<env assignments per run, ';' separated>"
set -uo pipefail
tag="${1:-s}"; out="gpurun_out/ab_$tag"; mkdir -p "$out"
cd "$GRAFT_REPO_ROOT"
quick="--no-cpu-baseline --no-bf16-secondary --no-north-star --no-configs --no-through-engine"
IFS=';' read -ra runs <<< "${2:-EP_INPASS=7}"
i=0
for r in "${runs[@]}"; do
  i=$((i+1))
  echo "=== $r"
  env $r EP_IP_STAMP=1 timeout 300 python bench.py --steps 60 --warmup 5 --spinup 10 $quick --kernel-iters 2 > "$out/stamp_$i.json" 2> "$out/stamp_$i.err"
  grep -A4 "EP_IP_STAMP. bwd2" "$out/stamp_$i.err" | tail -10
done
