"""End-to-end wall time of Optimizer.optimize() (diagnostic): the README example, a mid-size batch and
BASELINE config 3 through the plugin API, with and without the default pre-fusing."""
import pathlib
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import synthetic as syn  # noqa: E402
from tnco_amd.app import Optimizer  # noqa: E402

warnings.simplefilter("ignore")


def spec_of(n, seed):
    ts, _d, _ = syn.random_regular_tn(n, 3, seed)
    n_inds = max(max(x) for x in ts) + 1
    return [(2, *[f"t{t}" for t in range(n) if k in ts[t]]) for k in range(n_inds)]


def timed(label, spec, max_width=None, **kw):
    opt = Optimizer(method="sa", seed=0, max_width=max_width)
    t0 = time.perf_counter()
    tn, res = opt.optimize(spec, betas=(0, 100), **kw)
    dt = time.perf_counter() - t0
    c = float(res[0].cost)
    print(f"{label:46s} {dt:7.2f} s wall   {len(tn.tensors):4d} tensors after load   "
          f"best log2(cost) {np.log2(c) if c > 0 else float('-inf'):.3f}")


timed("README chain, 8 runs x 100 steps (warm-up)", "2 a b\n2 b c\n2 c d", n_steps=100, n_runs=8, fuse=None)
timed("README chain, 8 runs x 100 steps", "2 a b\n2 b c\n2 c d", n_steps=100, n_runs=8, fuse=None)
timed("README chain, default fuse", "2 a b\n2 b c\n2 c d", n_steps=100, n_runs=8)
timed("64 leaves, 4096 runs x 1000 steps, fuse=None", spec_of(64, 7), n_steps=1000, n_runs=4096, fuse=None)
timed("512 leaves, 65536 runs x 1000 steps, fuse=None", spec_of(512, 11), n_steps=1000, n_runs=65536, top_k=16, fuse=None)
timed("512 leaves, 65536 runs x 1000 steps, fuse=4", spec_of(512, 11), n_steps=1000, n_runs=65536, top_k=16)
timed("  same, fuse=None, initial_trees='kruskal'", spec_of(512, 11), n_steps=1000, n_runs=65536, top_k=16, fuse=None,
      initial_trees="kruskal")
timed("  same, fuse=None, top_k=1024 (default)", spec_of(512, 11), n_steps=1000, n_runs=65536, fuse=None)


def spec_of_cz(depth, fuse):
    """index-list spec of the CZ (hyper-index) circuit network: what the reference's loader makes of a circuit by default"""
    ts, _d, _o = syn.sycamore53_cz_tn(depth, fuse)
    n_inds = max(max(x) for x in ts if x) + 1
    return [(2, *[f"t{t}" for t in range(len(ts)) if k in ts[t]]) for k in range(n_inds)]


timed("CZ circuit depth 20 fused (hyper-indices), 65536 runs x 1000 steps", spec_of_cz(20, 4), n_steps=1000, n_runs=65536, top_k=16, fuse=None)
timed("  same once more", spec_of_cz(20, 4), n_steps=1000, n_runs=65536, top_k=16, fuse=None)
timed("CZ circuit depth 12 raw, 1000 tensors, 65536 runs x 1000 steps", spec_of_cz(12, None), n_steps=1000, n_runs=65536, top_k=16, fuse=None)
timed("CZ circuit depth 20 fused, max_width 42, 32768 runs x 1000 steps", spec_of_cz(20, 4), n_steps=1000, n_runs=32768, top_k=16, fuse=None, max_width=42)


def phases(n=512, R=65536, k=1024, sweeps=1000):
    """The pieces of such a call through the C ABI: initial trees, create, sweeps, read-back of the k best."""
    from tnco_amd import core
    prob = syn.regular_problem(n, 11)
    seeds = syn.replica_seeds(R)
    t = [time.perf_counter()]
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0); t.append(time.perf_counter())
    opt = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds); opt.sync(); t.append(time.perf_counter())
    betas = syn.linear_betas(0, 100, sweeps)
    for s in range(0, sweeps, 100):
        opt.run(betas[s:s + 100])
    opt.sync(); t.append(time.perf_counter())
    mn = opt.costs()[1]; t.append(time.perf_counter())
    c, ids = opt.best(k); t.append(time.perf_counter())
    lk, con = opt.trees(ids, which_min=True); t.append(time.perf_counter())
    paths = core.linear_paths(con, np.arange(n, dtype=np.int32), n); t.append(time.perf_counter())
    old = [opt.tree(int(r), which_min=True, with_masks=False) for r in ids[:64]]; t.append(time.perf_counter())
    d = np.diff(t)
    print(f"phases, {n} leaves x {R} runs x {sweeps} sweeps, head of {k}: greedy trees (device) {d[0]:.2f} s, create {d[1]:.2f} s, "
          f"sweeps {d[2]:.2f} s | read-back: all min costs {1e3 * d[3]:.1f} ms, k-select {1e3 * d[4]:.1f} ms, "
          f"{k} best trees + contractions {1e3 * d[5]:.1f} ms, {k} linear paths {1e3 * d[6]:.1f} ms "
          f"(round 1: one tree at a time {1e3 * d[7] / 64:.2f} ms each)")


phases()
