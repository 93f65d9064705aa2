"""cProfile of one optimize() call (65 536 runs x 1 000 sweeps, 512 leaves):  python tools/profile_optimize.py [top_k]"""
import sys, time, warnings, cProfile, pstats
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1])); warnings.simplefilter("ignore")
from tnco_amd import synthetic as syn
from tnco_amd.app import Optimizer
def spec_of(n, seed):
    ts, _d, _ = syn.random_regular_tn(n, 3, seed)
    n_inds = max(max(x) for x in ts) + 1
    return [(2, *[f"t{t}" for t in range(n) if k in ts[t]]) for k in range(n_inds)]
sp = spec_of(512, 11)
top_k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for i in range(2):
    Optimizer(method="sa", seed=0).optimize(sp, betas=(0, 100), n_steps=1000, n_runs=65536, top_k=top_k, fuse=None)
pr = cProfile.Profile(); pr.enable()
t0=time.perf_counter()
Optimizer(method="sa", seed=0).optimize(sp, betas=(0, 100), n_steps=1000, n_runs=65536, top_k=top_k, fuse=None)
print("wall", time.perf_counter()-t0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
