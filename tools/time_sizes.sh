#!/bin/bash
# tools/time_sizes.sh -- bench.py's headline leg on other network sizes / lane layouts (no PMC, no CPU leg)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for spec in "64 7 65536" "128 11 65536" "256 11 65536" "512 11 65536" "680 11 65536" "1024 11 65536" "1360 11 32768" "2048 11 32768"; do
  set -- $spec
  python3 "$ROOT/bench.py" --workload im --leaves $1 --graph-seed $2 --replicas $3 --steps ${STEPS:-8} --warmup 2 --pmc 0 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']
print('| %s leaves | bond dim 2, %d indices, %d mask words | %.2e |' % ('$1', c['n_inds'], c['mask_words'], j['value']))"
done
