#!/usr/bin/env bash
# Run ON THE GPU BOX: device durations (rocprofv3 kernel trace) of ONE planes contraction launched back to back, under
# several library settings.  usage: tools/exp_planes.sh OUTDIR "ENV1" "ENV2" ...   (OPS="logits dz" by default)
out="$1"; shift; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  for op in ${OPS:-logits dz}; do
    d="$out/$(echo "$cfg" | tr ' =' '__')_$op"
    env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/planes_one.py $op 40 > /dev/null 2> "$d.log"
    echo -n "$cfg $op: "; python3 tools/prof_summary.py "$d" 2>/dev/null | grep -i "planes_kernel\|gemm" | head -2 | awk '{printf "%s calls=%s avg_us=%s | ", $1, $(NF-3), $(NF-2)}'; echo
    rm -rf "$d"
  done
done
