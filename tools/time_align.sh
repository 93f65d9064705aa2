ROOT=$GRAFT_REPO_ROOT
for al in 0 128; do
for spec in "680 11 65536" "1024 11 65536" "1360 11 32768" "2048 11 32768"; do
  set -- $spec
  TNCO_HIP_BLOCK_ALIGN=$al python3 "$ROOT/bench.py" --workload im --leaves $1 --graph-seed $2 --replicas $3 --steps 3 --warmup 1 --pmc 0 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']['workload']
print('align $al | %s leaves | %s | %.3e |' % ('$1', c.split('(')[1].split(')')[0], j['value']))"
done
done
