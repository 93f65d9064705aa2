#!/bin/bash
# round 5, third probe: small batches spread over the chip (sa_run_kernel's spread) -- parity, the headline unchanged, the
# latency regime with and without
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests -m gpu -x -q < /dev/null > "$OUT/gputest_c.log" 2>&1
tail -3 "$OUT/gputest_c.log"
for i in 1 2; do
timeout 300 python bench.py --workload im --pmc 0 --cpu-sample 0 --e2e 0 --steps 12 --warmup 3 < /dev/null 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline %.4e  ms/step %.2f bad %s' % (j['value'], j['ms_per_step'], j['config']['validated_bad_replicas']))"
done
echo "## spread on (default)"; timeout 500 python tools/latency_regime.py --cpu-budget 0.001 < /dev/null 2>&1 | tee "$OUT/latency_spread_on.txt"
echo "## TNCO_HIP_SPREAD=0"; TNCO_HIP_SPREAD=0 timeout 500 python tools/latency_regime.py --cpu-budget 0.001 < /dev/null 2>&1 | tee "$OUT/latency_spread_off.txt"
