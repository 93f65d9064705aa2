#!/bin/bash
# tools/r04_measure.sh -- the round-4 measurement set on the GPU box, from the library in the tree:
#   gpurun_out/r04/{bench.json,pmc_traffic.json,kernel_stats.csv}   tools/profile_round.sh (bench + PMC passes + kernel trace)
#   gpurun_out/r04/fw_widths.txt    the finite-width leg of bench.py at max_width 28 / 32 / 40 (+ the easier network of rounds 1-3)
#   gpurun_out/r04/fw_cliff.txt     tools/fw_widths.py on config 5, two hyper-index circuit networks, a 1000-tensor network
#   gpurun_out/r04/e2e.txt          tools/time_e2e.py
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 bash tools/profile_round.sh r04 > "$OUT/profile_round.log" 2>&1
cp gpurun_out/prof_r04/bench.json gpurun_out/prof_r04/pmc_traffic.json gpurun_out/prof_r04/kernel_stats.csv "$OUT/" 2>/dev/null
tail -12 "$OUT/profile_round.log"
: > "$OUT/fw_widths.txt"
for cfg in "supremacy 28" "supremacy 32" "supremacy 40" "alternating 40"; do
  set -- $cfg
  timeout 300 python bench.py --workload fw --fw-layout $1 --fw-max-width $2 --pmc 0 --cpu-sample 0 --steps 20 --warmup 5 > /tmp/fw.json 2>/dev/null
  python - "$1" "$2" >> "$OUT/fw_widths.txt" <<'PY'
import json, sys
j = json.load(open("/tmp/fw.json"))
r = j["roofline"]
print(f"{sys.argv[1]:12s} max_width {sys.argv[2]:>3s}: {j['value']:.3e} move-evals/s  {j['ms_per_step']:.2f} ms/step  "
      f"moves {r['kernels']['fw_move_kernel']['ms_per_step']:.2f} ms  re-slice {r['kernels']['fw_reslice_kernel']['ms_per_step']:.2f} ms (per stream)  "
      f"accept {j['config']['accept_rate']:.3f}  best log10(flops) {j['config']['best_log10_flops']:.2f}  "
      f"left the one-wavefront path {r['reslices']['left_one_wavefront_path_frac']:.2e}")
PY
done
cat "$OUT/fw_widths.txt"
: > "$OUT/fw_cliff.txt"
for net in sycamore cz:20:4 cz:12:raw regular:1000; do
  timeout 600 python tools/fw_widths.py --network $net --frac 0.7 --replicas 32768 --sweeps 300 --chunk 100 2>&1 | cut -c1-330 >> "$OUT/fw_cliff.txt"
done
grep -c sweeps "$OUT/fw_cliff.txt"
timeout 600 python tools/time_e2e.py > "$OUT/e2e.txt" 2>&1
tail -4 "$OUT/e2e.txt"
