for i in 1 2; do
for lib in build_variants/lib_maxd64.so build_variants/lib_maxd80.so tnco_amd/libtnco_hip.so; do
for cfg in "supremacy 40" "alternating 40"; do set -- $cfg
TNCO_HIP_LIB=$PWD/$lib timeout 300 python bench.py --workload fw --fw-layout $1 --fw-max-width $2 --pmc 0 --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib $1 $2', round(j['value']/1e9,3), round(j['ms_per_step'],2))"
done; done; done
