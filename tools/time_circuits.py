"""The infinite-memory sweep kernel on the circuit networks (65 536 replicas, greedy starts, six calls of 100 sweeps back
to back after two warm-up calls): the Sycamore-53 supremacy sequence and the two CZ-decomposed networks whose
hyper-indices are what the reference's loader produces by default (tnco/app/app.py:351-358)."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import core, synthetic  # noqa: E402


def run(label, p, R=65536, S=100, K=6):
    seeds = synthetic.replica_seeds(R)
    t0 = time.perf_counter()
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, output_mask=p.output_mask, device=0, keep_on_device=True)
    tg = time.perf_counter() - t0
    with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, output_mask=p.output_mask) as opt:
        betas = synthetic.linear_betas(0.0, 100.0, S * (K + 2))
        opt.run(betas[:S]); opt.run(betas[S:2 * S]); opt.sync()
        m0 = opt.counters()["moves"]
        t0 = time.perf_counter()
        for k in range(2, K + 2):
            opt.run(betas[k * S:(k + 1) * S])
        opt.sync()
        dt = time.perf_counter() - t0
        mv = opt.counters()["moves"] - m0
        bad = opt.validate()[0]
        print(f"{label:84s} {mv / dt:9.3e} move-evals/s   best log10(flops) {np.log10(opt.costs()[1].min()):6.2f} after {S * (K + 2)} sweeps   "
              f"bad replicas {bad}   initial trees {1e3 * tg:.0f} ms", flush=True)


p = synthetic.sycamore_problem(20, "supremacy")
run(f"Sycamore-53 supremacy sequence, {p.n} tensors, {p.W} mask words", p)
ts, dims, out = synthetic.sycamore53_cz_tn(20, 4)
p = synthetic.Problem(ts, 2, out)
run(f"CZ circuit depth 20, fused (loader default): {p.n} tensors, {p.W} words, hyper-indices", p)
ts, dims, out = synthetic.sycamore53_cz_tn(12, None)
p = synthetic.Problem(ts, 2, out)
run(f"CZ circuit depth 12, raw: {p.n} tensors, {p.W} words, hyper-indices", p)
