"""optimize() of 65 536 runs x 1 000 sweeps on the 512-leaf network, several times, and the same call in its
pieces through the C ABI (diagnostic; TNCO_HIP_DEBUG=1 print the steps inside):
    python tools/e2e_debug.py [repeats]"""
import pathlib
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
warnings.simplefilter("ignore")
from tnco_amd import core, synthetic as syn  # noqa: E402
from tnco_amd.app import Optimizer  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def spec_of(n, seed):
    ts, _d, _ = syn.random_regular_tn(n, 3, seed)
    n_inds = max(max(x) for x in ts) + 1
    return [(2, *[f"t{t}" for t in range(n) if k in ts[t]]) for k in range(n_inds)]


sp = spec_of(512, 11)


def _timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            if dt > 0.03:
                print(f"    {label} took {1e3 * dt:.0f} ms")
    setattr(obj, name, g)


_timed(core, "greedy_release", "greedy_release()")
_timed(core, "greedy_trees", "greedy_trees()")
_timed(core.BatchedOptimizer, "close", "BatchedOptimizer.close()")
_timed(core.BatchedOptimizer, "__init__", "BatchedOptimizer()")
_timed(core.BatchedOptimizer, "costs", "costs() [waits for the sweeps]")
for i in range(reps):
    opt = Optimizer(method="sa", seed=0)
    t0 = time.perf_counter()
    tn, res = opt.optimize(sp, betas=(0, 100), n_steps=1000, n_runs=65536, top_k=16, fuse=None)
    print(f"optimize() {time.perf_counter() - t0:.3f} s")

prob = syn.regular_problem(512, 11)
seeds = syn.replica_seeds(65536)
betas = syn.linear_betas(0, 100, 1000)
for i in range(reps):
    t = [time.perf_counter()]
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0, keep_on_device=True); t.append(time.perf_counter())
    opt = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds); opt.sync(); t.append(time.perf_counter())
    for s in range(0, 1000, 100):
        opt.run(betas[s:s + 100])
    opt.sync(); t.append(time.perf_counter())
    c, ids = opt.best(16); lk, con = opt.trees(ids, which_min=True); t.append(time.perf_counter())
    opt.close(); t.append(time.perf_counter())
    d = 1e3 * np.diff(t)
    print(f"pieces: trees {d[0]:.1f} ms, create {d[1]:.1f} ms, sweeps {d[2]:.1f} ms, 16 best {d[3]:.1f} ms, destroy {d[4]:.1f} ms")
