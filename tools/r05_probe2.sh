#!/bin/bash
# round 5, second probe: the split + copy layout of the infinite-memory networks beyond 12 mask words -- parity tests, then
# A/B against the packed layout (TNCO_HIP_BLOCK_ALIGN=-1) on the circuit networks and the larger 3-regular ones
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests -m gpu -x -q < /dev/null > "$OUT/gputest_b.log" 2>&1
tail -4 "$OUT/gputest_b.log"
: > "$OUT/split_ab.txt"
for v in 0 -1; do
  echo "## TNCO_HIP_BLOCK_ALIGN=$v TNCO_HIP_HYPER_ALIGNED=$((v+1))" >> "$OUT/split_ab.txt"
  TNCO_HIP_BLOCK_ALIGN=$v TNCO_HIP_HYPER_ALIGNED=$((v+1)) timeout 300 python tools/time_circuits.py < /dev/null >> "$OUT/split_ab.txt" 2>&1
  for spec in "680 11 65536" "1024 11 65536" "1360 11 32768" "2048 11 32768"; do
    set -- $spec
    TNCO_HIP_BLOCK_ALIGN=$v timeout 200 python bench.py --workload im --leaves $1 --graph-seed $2 --replicas $3 --steps 6 --warmup 2 --pmc 0 --cpu-sample 0 --e2e 0 < /dev/null 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']
print('| %s leaves | %d mask words | %.2e | bad %s |' % ('$1', c['mask_words'], j['value'], c['validated_bad_replicas']))" >> "$OUT/split_ab.txt"
  done
done
cat "$OUT/split_ab.txt"
