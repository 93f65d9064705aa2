#!/bin/bash
# tools/wide_pmc.sh -- the headline kernel on networks beyond 12 mask words with this run's PMC passes: fabric requests and
# bytes per move, request rate against the box's random-line rate -> gpurun_out/r06/wide_pmc.txt (profiles/r06_wide_pmc.txt)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06; mkdir -p "$OUT"; cd "$ROOT"
: > "$OUT/wide_pmc.txt"
for spec in "512 65536" "680 65536" "1024 65536" "2048 32768"; do
  set -- $spec
  timeout 600 python bench.py --workload im --leaves $1 --graph-seed 11 --replicas $2 --steps 8 --warmup 2 --cpu-sample 0 --e2e 0 < /dev/null 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; c=j['config']; q=r.get('requests_per_move') or {}; t=r.get('traffic_per_move') or {}
print('| %s leaves | %d mask words | %s | %.3e move-evals/s | requests/move read %.2f write %.2f | bytes/move read %.0f write %.0f | frac %.3f | request rate %.3e/s = %.2f of this box\'s random-line rate | accept %.2f |' % ('$1', c['mask_words'], r['kernel'], j['value'], q.get('read',0), q.get('write',0), t.get('read',0), t.get('write',0), r['frac'] or 0, r.get('request_rate',0), r.get('request_rate_frac_this_box',0), c['accept_rate']))" | tee -a "$OUT/wide_pmc.txt"
done
