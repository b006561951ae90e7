#!/usr/bin/env bash
# usage: tools/ab_variants.sh "<microbench args>" tag1 tag2 ...   ("base" = the shipped library)
args="$1"; shift
for round in 1 2; do
for tag in "$@"; do
  if [[ "$tag" == base ]]; then lib=""; else lib="$PWD/efficient_probing_amd/variants/libep_hip_$tag.so"; fi
  for ab in 0 1; do
    echo -n "$tag ablate=$ab: "; EP_HIP_LIB="$lib" EP_POOL_ABLATE=$ab python tools/pool_microbench.py $args 2>/dev/null | cut -c1-110
  done
done
done
