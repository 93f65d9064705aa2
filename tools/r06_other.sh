#!/bin/bash
# round 6: throughput outside the two bench legs -> gpurun_out/r06/other_configs.txt (profiles/r06_other_configs.md)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
{
echo "## tools/time_models.py"; timeout 400 python tools/time_models.py < /dev/null 2>&1
echo "## tools/time_circuits.py"; timeout 400 python tools/time_circuits.py < /dev/null 2>&1
echo "## tools/time_sizes.sh"; STEPS=6 timeout 600 bash tools/time_sizes.sh < /dev/null 2>&1
} > "$OUT/other_configs.txt"
cat "$OUT/other_configs.txt"
