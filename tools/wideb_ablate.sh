export EP_POOL_WIDEB=1
for v in "$@"; do echo "== $v"; if [ "$v" = full ]; then timeout 100 python tools/wideb_check.py 2>&1 | grep "ward pass"; elif [ "$v" = old ]; then EP_POOL_WIDEB=0 timeout 100 python tools/wideb_check.py 2>&1 | grep "ward pass"; else EP_HIP_LIB=$PWD/efficient_probing_amd/variants/libep_hip_$v.so timeout 100 python tools/wideb_check.py 2>&1 | grep "ward pass"; fi; done
