#!/bin/bash
# round 5, first probe set: the two-line hyper layout (tests, A/B against the packed one, request counters), the
# proposal == current statistics of the finite-width re-slice, the latency regime table
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_app.py tests/test_gpu_extract.py tests/test_gpu_restore.py -x -q -k "hyper or vector or random or cz or component or golden" < /dev/null > "$OUT/hyper_tests.log" 2>&1
tail -4 "$OUT/hyper_tests.log"
for v in 1 0; do
  echo "TNCO_HIP_HYPER_ALIGNED=$v" >> "$OUT/hyper_ab.txt"
  TNCO_HIP_HYPER_ALIGNED=$v timeout 200 python tools/time_models.py --only hyper < /dev/null >> "$OUT/hyper_ab.txt" 2>&1
  TNCO_HIP_HYPER_ALIGNED=$v timeout 200 python tools/time_models.py --only hyper < /dev/null >> "$OUT/hyper_ab.txt" 2>&1
done
cat "$OUT/hyper_ab.txt"
PROF_SCRIPT=tools/time_models.py timeout 600 bash tools/profile_mem.sh r05_hyper --only hyper < /dev/null > "$OUT/hyper_pmc.txt" 2>&1
tail -30 "$OUT/hyper_pmc.txt"
timeout 400 python tools/fw_changed_hist.py 32 65536 < /dev/null > "$OUT/fw_changed_hist.txt" 2>&1
tail -22 "$OUT/fw_changed_hist.txt"
timeout 600 python tools/latency_regime.py < /dev/null > "$OUT/latency_regime.txt" 2>&1
cat "$OUT/latency_regime.txt"
