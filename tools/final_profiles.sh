#!/usr/bin/env bash
# Run ON THE GPU BOX: the round's profile set (tools/make_profiles.sh + tools/mfma_util.sh per tag); PART=1|2|3 picks a third.
part="${PART:-1}"
run() { tag="$1"; shift; bash tools/make_profiles.sh "$tag" "$@" > /dev/null 2>&1; bash tools/mfma_util.sh "$tag" --no-configs "$@" > /dev/null 2>&1; echo "== $tag"; head -4 "gpurun_out/profiles_$tag/kernel_stats_summary.txt"; }
if [[ "$part" == 1 ]]; then
  run c2
  run ns --workload ns
  run c2_bf16 --tokens bf16
  run c2_bf16_amp --tokens bf16 --arith bf16_autocast
elif [[ "$part" == 2 ]]; then
  run c5 --workload c5
  run c5_bf16 --workload c5 --tokens bf16
  run c2_q32 --queries 32
  run c2_q32_bf16 --queries 32 --tokens bf16
else
  run c3_q32 --workload c3 --queries 32
  run c4_q32 --workload c4 --queries 32
  EP_PROF_STEPS="--steps 10 --warmup 3" EP_PROF_TIMELINE_STEP=8 run c4_abmilp --head abmilp --workload c4
  EP_PROF_STEPS="--steps 10 --warmup 3" EP_PROF_TIMELINE_STEP=8 run c4_abmilp_amp --head abmilp --workload c4 --arith bf16_autocast
fi
