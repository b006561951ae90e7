#!/usr/bin/env bash
# Same-box A/B of bench.py under environment switches.  usage (via gpurun): bash tools/ab_bench.sh "<bench args>" "ENV_A" "ENV_B" ... [REPS=2]
# Each arm is run REPS times, alternating; prints ms_per_step and img/s per run.
set -uo pipefail
bargs="$1"; shift
reps="${REPS:-2}"
quick="--no-cpu-baseline --no-bf16-secondary --no-north-star --no-configs --no-through-engine --kernel-iters 3"
for r in $(seq 1 $reps); do
  for e in "$@"; do
    out=$(env $e python3 bench.py $quick $bargs 2>/dev/null | tail -1)
    echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; i=r.get('in_step',{}); print('AB', '$e', '| $bargs |', d['ms_per_step'], d['value'], 'fwd', i.get('fwd_us'), 'bwd', i.get('bwd_us'))" || echo "AB $e FAILED: $out"
  done
done
