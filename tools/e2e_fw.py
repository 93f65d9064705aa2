import sys, time, warnings
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1])); warnings.simplefilter("ignore")
import numpy as np
from tnco_amd import core, synthetic as syn
from tnco_amd.app import Optimizer
ts, _d, out = syn.sycamore53_tn(20)
n_inds = max(max(x) for x in ts) + 1
spec = [(2, *[f"t{t}" for t in range(len(ts)) if k in ts[t]]) for k in range(n_inds)]
for i in range(4):
    opt = Optimizer(method="sa", seed=0, max_width=32)
    t0 = time.perf_counter()
    tn, res = opt.optimize(spec, betas=(0, 100), n_steps=1000, n_runs=65536, top_k=16, fuse=None)
    print(f"optimize(max_width=32) {time.perf_counter() - t0:.3f} s   best {float(res[0].cost):.4g}", file=sys.stderr)
