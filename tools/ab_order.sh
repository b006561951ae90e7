#!/usr/bin/env bash
# Run ON THE GPU BOX: A/B of an environment switch over a list of bench workloads (alternating, same box).
# usage: OLD="EP_B3_XCD=0" bash tools/ab_order.sh   (STEPS, REPS optional)
common="--no-cpu-baseline --no-north-star --no-bf16-secondary --no-configs --no-through-engine --steps ${STEPS:-30} --warmup 5"
run() { python3 bench.py $common "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8.4f ms' % d['ms_per_step'])"; }
IFS=';' read -ra CFGS <<< "${CFGS:---head abmilp --workload c4;--head abmilp --workload c4 --arith bf16_autocast;--head abmilp;--head dinovit;--head dolg;--head coca --workload c4;--workload c3;--workload c4;--workload c5;--workload c2 --queries 32;--workload c4 --queries 32;--workload c5 --arith bf16_autocast}"
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "${CFGS[@]}"; do
    a=$(env $OLD bash -c "$(declare -f run); common='$common'; run $cfg"); b=$(run $cfg)
    echo "$cfg : old $a   new $b"
  done
done
