export EP_PLANES_BIG=1 PROBE_SHAPES=${PROBE_SHAPES:-1024x4096x4096,1024x4096x1000}
for v in "$@"; do echo "== $v"; if [ "$v" = full ]; then timeout 100 python tools/planes_probe.py 2>&1 | grep " x "; else EP_HIP_LIB=$PWD/efficient_probing_amd/variants/libep_hip_pbg_$v.so timeout 100 python tools/planes_probe.py 2>&1 | grep " x "; fi; done
