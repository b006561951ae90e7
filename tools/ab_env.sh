#!/usr/bin/env bash
# Same-box A/B of bench.py over environment switches (optionally a pytest run in front): one gpurun call, fresh processes.
# usage (via gpurun): bash tools/ab_env.sh TAG "<env sets ';' separated>" "<workloads>" ["<pytest env>" pytest targets...]
set -uo pipefail
tag="${1:-x}"; IFS=';' read -ra runs <<< "${2:-EP_INPASS=0}"; wls="${3:-c2}"; shift 3 || true
out="gpurun_out/ab_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$#" -gt 1 ]; then
  penv="$1"; shift
  env $penv timeout 1500 python -m pytest "$@" -x -q > "$out/tests.log" 2>&1
  echo "tests ($penv) rc $?"; tail -3 "$out/tests.log"
fi
quick="--no-cpu-baseline --no-bf16-secondary --no-north-star --no-configs --no-through-engine"
for wl in $wls; do
  i=0
  for r in "${runs[@]}"; do
    i=$((i+1))
    env $r timeout 300 python bench.py --steps 100 --warmup 10 $quick --kernel-iters 5 --workload $wl > "$out/b_${wl}_$i.json" 2> "$out/b_${wl}_$i.err"
    python3 - "$out/b_${wl}_$i.json" "$wl [$r]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"].get("in_step", {})
    print(sys.argv[2], "ms/step", d["ms_per_step"], "p50", d.get("step_ms_p50"), "in-step fwd/bwd us", r.get("fwd_us"), r.get("bwd_us"), "loss", d["check"]["mean_loss_over_timed_steps"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
