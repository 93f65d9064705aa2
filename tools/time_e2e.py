"""End-to-end wall time of Optimizer.optimize() for a big batch (diagnostic)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tnco_amd import core, synthetic as syn
from tnco_amd.app import Optimizer
n, R = 512, 65536
ts, d, _ = syn.random_regular_tn(n, 3, 11)
I = max(max(x) for x in ts) + 1
spec = [(2, *[f"t{t}" for t in range(n) if k in ts[t]]) for k in range(I)]
t0 = time.perf_counter()
seeds = list(range(R))
links = core.random_trees(ts, I, seeds)
t1 = time.perf_counter()
print(f"random_trees: {t1-t0:.2f} s for {R} trees")
opt = Optimizer(method='sa', seed=0)
t0 = time.perf_counter()
tn, res = opt.optimize(spec, betas=(0, 100), n_steps=1000, n_runs=R, top_k=16)
t1 = time.perf_counter()
print(f"optimize(): {t1-t0:.2f} s wall, best log2 cost {np.log2(float(res[0].cost)):.3f}, runtime_s field {res[0].runtime_s:.2f}")
