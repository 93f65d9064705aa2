#!/bin/bash
# tools/ab_fw.sh VARIANT [N] -- the finite-width leg of bench.py, the library in the tree against build_variants/lib_VARIANT.so,
# alternating, N times each (same box, same call)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for i in $(seq ${2:-3}); do
  for lib in tree $1; do
    if [ $lib = tree ]; then unset TNCO_HIP_LIB; else export TNCO_HIP_LIB=$ROOT/build_variants/lib_$lib.so; fi
    timeout 300 python bench.py --workload fw --pmc 0 --cpu-sample 0 --e2e 0 --steps 20 --warmup 5 < /dev/null 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-10s %.4e move-evals/s  %.2f ms/step  moves %.2f  re-slice %.2f ms (per stream)  bad %s' % ('$lib', j['value'], j['ms_per_step'], r['kernels']['fw_move_kernel']['ms_per_step'], r['kernels']['fw_reslice_kernel']['ms_per_step'], j['config']['validated_bad_replicas']))"
  done
done
