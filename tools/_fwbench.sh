mkdir -p gpurun_out/r4c
for mw in 28 32 40; do
  python bench.py --workload fw --fw-max-width $mw --pmc 0 --cpu-sample 0 --steps 10 --warmup 2 > gpurun_out/r4c/fw_$mw.json 2> gpurun_out/r4c/fw_$mw.err
  python - <<PY
import json
j=json.load(open("gpurun_out/r4c/fw_$mw.json"))
r=j["roofline"]
print($mw, "%.3e"%j["value"], "ms/step %.2f"%j["ms_per_step"], {k:(round(v["ms_per_step"],3), v["launches_per_step"]) for k,v in r["kernels"].items()}, "best", j["config"]["best_log10_flops"], "acc", round(j["config"]["accept_rate"],3), "moves", j["config"]["moves_timed"])
PY
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4c/prof32 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload fw --fw-max-width 32 --pmc 0 --cpu-sample 0 --steps 10 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r4c/prof32 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -12'
find gpurun_out/r4c/prof32 -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r4c/prof32 -name "*.db" -delete
