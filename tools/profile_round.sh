#!/bin/bash
# tools/profile_round.sh TAG [bench args...] -- the evidence behind the bench line, from the library
# in the tree, on the GPU box:
#   1. bench.py itself (its own rocprofv3 --pmc child passes give roofline.traffic / request counts;
#      the per-step counters also go to pmc_traffic.json),
#   2. rocprofv3 --kernel-trace --stats of the same command (kernel_stats.csv).
# Output under gpurun_out/prof_<TAG>/; copy bench.json, pmc_traffic.json and kernel_stats.csv into
# profiles/ (<TAG>_bench.json, pmc_traffic.json, <TAG>_kernel_stats.csv).
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" "$@" --pmc-out "$OUT/pmc_traffic.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc $?"; tail -c 600 "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" "$@" --pmc 0 --cpu-sample 0 > "$OUT/trace.log" 2>&1
echo "trace rc $?"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
head -8 "$OUT/kernel_stats.csv"
python3 - "$OUT/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(name, o):
    r = o["roofline"]
    print(f"{name}: {o['value']:.4g} {o['unit']}  ms/step {o['ms_per_step']:.2f}  frac {r['frac']:.3f} "
          f"traffic_frac {r.get('traffic_frac')}  req/move {r.get('requests_per_move')}  req_rate_frac {r.get('request_rate_frac')}")
    for k, v in r["kernels"].items():
        print("   ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
    if "cpu_baseline" in o:
        print("    cpu", o["cpu_baseline"]["value"], o["cpu_baseline"]["cores"], o["config"].get("cpu_sample_min_cost_bit_exact"))
show("im", j)
if "fw" in j: show("fw", j["fw"])
PY
