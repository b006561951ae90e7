#!/usr/bin/env bash
# Copy the summaries tools/make_profiles.sh / tools/mfma_util.sh left under gpurun_out/ into profiles/RND/ (tracked).
# usage: tools/collect_profiles.sh r05 c2 ns c5 ...
set -euo pipefail
rnd="$1"; shift
dst="profiles/$rnd"; mkdir -p "$dst"
for tag in "$@"; do
  src="gpurun_out/profiles_$tag"
  if [[ -d "$src" ]]; then
    cp "$src/bench_under_trace.json" "$dst/${tag}_bench_under_trace.json"
    cp "$src/kernel_stats_summary.txt" "$dst/${tag}_kernel_stats_summary.txt"
    cp "$src/step_timeline.txt" "$dst/${tag}_step_timeline.txt"
    cp "$src/hbm_traffic.json" "$dst/${tag}_hbm_traffic_pmc.json"
    f=$(find "$src/trace" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" "$dst/${tag}_rocprofv3_kernel_stats.csv"
  fi
  m="gpurun_out/mfma_$tag/mfma_util.json"
  [[ -f "$m" ]] && cp "$m" "$dst/${tag}_mfma_util_pmc.json"
done
ls "$dst" | wc -l
