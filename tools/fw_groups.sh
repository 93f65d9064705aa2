#!/bin/bash
# tools/fw_groups.sh -- the finite-width leg with the batch split over 1 / 2 / 3 / 4 streams (TNCO_HIP_GROUPS; the library's choice is 2)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for G in 2 1 3 4 2; do
  TNCO_HIP_GROUPS=$G timeout 300 python bench.py --workload fw --pmc 0 --cpu-sample 0 --e2e 0 --calibrate 0 --steps 20 --warmup 5 < /dev/null 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('streams $G: %.4e move-evals/s  %.2f ms/step  moves %.2f  re-slice %.2f ms (per stream)  bad %s' % (j['value'], j['ms_per_step'], r['kernels']['fw_move_kernel']['ms_per_step'], r['kernels']['fw_reslice_kernel']['ms_per_step'], j['config']['validated_bad_replicas']))"
done
