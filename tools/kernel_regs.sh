#!/usr/bin/env bash
# Per-kernel register / LDS / spill figures of one built object (no GPU needed):
#   tools/kernel_regs.sh efficient_probing_amd/csrc/ep_pool_stream.o [name-filter]
# Also leaves the gfx950 code object at /tmp/kr_<name>.co for llvm-objdump -d.
set -euo pipefail
B=/opt/rocm/lib/llvm/bin
obj="$1"; filt="${2:-}"
n=$(basename "$obj" .o)
$B/llvm-objcopy --dump-section .hip_fatbin=/tmp/kr_$n.fat "$obj"
$B/clang-offload-bundler --unbundle --input=/tmp/kr_$n.fat --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kr_$n.co
$B/llvm-readelf --notes /tmp/kr_$n.co | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
filt = '$filt'
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'^void ', '', name)
    if filt and filt not in name: continue
    print(f'{name[:90]:90s} vgpr {g(\"vgpr_count\"):>4} sgpr {g(\"sgpr_count\"):>4} spill {g(\"vgpr_spill_count\"):>3} lds {g(\"group_segment_fixed_size\"):>6} scratch {g(\"private_segment_fixed_size\"):>4}')
"
