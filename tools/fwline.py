"""One line per bench.py JSON line on stdin: value, ms per step, per-kernel ms per step (experiments):
    python bench.py --workload fw --pmc 0 --cpu-sample 0 | python tools/fwline.py"""
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        k = d["roofline"]["kernels"]
        print("value %.4g  ms/step %.2f  " % (d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % (n.replace("_kernel", ""), v["ms_per_step"]) for n, v in k.items()))
