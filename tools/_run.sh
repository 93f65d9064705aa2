timeout 1200 python -m pytest tests/test_gpu_fw.py tests/test_gpu_restore.py tests/test_gpu_app.py -x -q 2>&1 | tail -5
for g in 1 2; do
  TNCO_HIP_GROUPS=$g timeout 200 python bench.py --workload fw --pmc 0 --cpu-sample 0 --steps 10 --warmup 2 > /tmp/fw_$g.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("/tmp/fw_$g.json"))
r=j["roofline"]
print("groups", $g, "%.3e"%j["value"], "ms/step %.2f"%j["ms_per_step"], {k:(round(v["ms_per_step"],3)) for k,v in r["kernels"].items()}, j["config"]["best_log10_flops"])
PY
done
