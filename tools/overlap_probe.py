#!/usr/bin/env python3
"""overlap_probe.py -- does splitting the replicas of the infinite-memory leg over G handles (G streams)
remove the launch tail of `sa_run_kernel`?  65536 replicas are 1024 blocks for 768 resident ones: the last
256 blocks of every launch run on a third of the chip.  With G handles of R / G replicas, each on a stream of
its own and stepped round-robin, a block that ends frees its slot for a block of ANOTHER handle's pending
launch, so the chip stays full over the whole schedule (experiment of round 3; the library does this
inside one handle when it pays).

    python tools/overlap_probe.py [--groups 1,2,4,8] [--steps 20] [--sweeps 100]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tnco_amd import core, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--replicas", type=int, default=65536)
    ap.add_argument("--leaves", type=int, default=512)
    a = ap.parse_args()
    prob = synthetic.regular_problem(a.leaves, graph_seed=11)
    seeds = synthetic.replica_seeds(a.replicas, S=0)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0)
    betas = synthetic.linear_betas(0.0, 100.0, (a.warmup + a.steps) * a.sweeps)
    for G in [int(x) for x in a.groups.split(",")]:
        per = a.replicas // G
        opts = [core.BatchedOptimizer(prob.leaf_masks, links[g * per:(g + 1) * per], seeds[g * per:(g + 1) * per],
                                      n_inds=prob.n_inds, dims=2, device=0) for g in range(G)]
        for s in range(a.warmup):
            for o in opts:
                o.run(betas[s * a.sweeps:(s + 1) * a.sweeps])
        for o in opts:
            o.sync()
        m0 = sum(o.counters()["moves"] for o in opts)
        t0 = time.perf_counter()
        for s in range(a.warmup, a.warmup + a.steps):
            for o in opts:
                o.run(betas[s * a.sweeps:(s + 1) * a.sweeps])
        for o in opts:
            o.sync()
        dt = time.perf_counter() - t0
        m1 = sum(o.counters()["moves"] for o in opts)
        best = min(float(o.costs()[1].min()) for o in opts)
        print(f"groups {G}: {(m1 - m0) / dt / 1e9:.3f}e9 move-evals/s, {dt / a.steps * 1e3:.2f} ms per step, "
              f"best log10 {np.log10(best):.4f}", flush=True)
        for o in opts:
            o.close()


if __name__ == "__main__":
    main()
