for i in 1 2 3; do
for lib in build_variants/lib_maxd64.so tnco_amd/libtnco_hip.so; do
TNCO_HIP_LIB=$PWD/$lib timeout 300 python bench.py --workload fw --pmc 0 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib', round(j['value']/1e9,3), round(j['ms_per_step'],2), r['reslices']['fell_back'])"
done; done
