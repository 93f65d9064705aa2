#!/bin/bash
# tools/r06_measure.sh PART -- the round-6 measurement set on the GPU box, from the library in the tree, in parts
# (one gpurun call each; everything lands in gpurun_out/r06/ and is copied into profiles/r06_*):
#   bench     tools/profile_round.sh: bench.py --gpus 1 --steps 20 --warmup 5 (the command the driver runs) with its PMC passes + rocprofv3 --kernel-trace --stats of the same command
#   other     tools/r06_other.sh: cost models, circuit networks, network sizes
#   widths    the finite-width leg at max_width 28 / 32 / 40
#   e2e       tools/time_e2e.py, tools/latency_regime.py
#   validate  tools/validate_round.sh (2 500-sweep runs with a CPU sample; 524 288 replicas on one GPU)
#   fuzz      tools/fuzz_round.sh
set -u
PART=${1:-bench}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
case $PART in
bench)
  timeout 1200 bash tools/profile_round.sh r06 --gpus 1 --steps 20 --warmup 5 < /dev/null > "$OUT/profile_round.log" 2>&1
  cp gpurun_out/prof_r06/bench.json gpurun_out/prof_r06/pmc_traffic.json gpurun_out/prof_r06/kernel_stats.csv "$OUT/" 2>/dev/null
  tail -14 "$OUT/profile_round.log" ;;
other)
  bash tools/r06_other.sh < /dev/null ;;
widths)
  : > "$OUT/fw_widths.txt"
  for W in 28 32 40; do
    timeout 300 python bench.py --workload fw --fw-max-width $W --pmc 0 --cpu-sample 0 --e2e 0 --steps 20 --warmup 5 < /dev/null 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; c=j['config']
print('max_width $W: %.3e move-evals/s  %.2f ms/step  moves %.2f ms  re-slice %.2f ms (per stream)  accept %.3f  best log10(flops) %.2f  bad replicas %s  left the one-wavefront path %.2e' % (j['value'], j['ms_per_step'], r['kernels']['fw_move_kernel']['ms_per_step'], r['kernels']['fw_reslice_kernel']['ms_per_step'], c['accept_rate'], c['best_log10_flops'], c['validated_bad_replicas'], r['reslices']['left_one_wavefront_path_frac']))" >> "$OUT/fw_widths.txt"
  done
  cat "$OUT/fw_widths.txt" ;;
e2e)
  timeout 600 python tools/time_e2e.py < /dev/null > "$OUT/e2e.txt" 2>&1; tail -12 "$OUT/e2e.txt"
  timeout 900 python tools/latency_regime.py < /dev/null > "$OUT/latency_regime.txt" 2>&1; cat "$OUT/latency_regime.txt" ;;
validate)
  timeout 1500 bash tools/validate_round.sh < /dev/null > "$OUT/validation.txt" 2>&1; cat "$OUT/validation.txt" ;;
fuzz)
  timeout 3000 bash tools/fuzz_round.sh < /dev/null > "$OUT/fuzz.txt" 2>&1; cat "$OUT/fuzz.txt" ;;
esac
