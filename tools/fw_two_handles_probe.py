#!/usr/bin/env python3
"""fw_two_handles_probe.py -- do two finite-width handles of half the replicas each, on streams of their own and out
of phase, overlap the request-bound moves of one with the latency-bound re-slice kernels of the other?  (Round 2
measured 60.3 against 64.5 ms per 100 sweeps with the old re-slice kernels.)

    python tools/fw_two_handles_probe.py [--steps 12] [--sweeps 100] [--offset 5]
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
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--replicas", type=int, default=65536)
    ap.add_argument("--offset", type=int, default=5, help="sweeps the second handle runs ahead (phase shift of its re-slices)")
    a = ap.parse_args()
    p = synthetic.sycamore_problem(20)
    seeds = synthetic.replica_seeds(a.replicas, S=0)
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, device=0)
    betas = synthetic.linear_betas(0.0, 100.0, (a.steps + 3) * a.sweeps)
    for G in (1, 2):
        per = a.replicas // G
        opts = [core.BatchedOptimizer(p.leaf_masks, links[g * per:(g + 1) * per], seeds[g * per:(g + 1) * per], n_inds=p.n_inds,
                                      dims=2, device=0, max_width=40.0) for g in range(G)]
        pos = [0] * G
        for g, o in enumerate(opts):  # warm-up, the second handle a few sweeps ahead
            n = 2 * a.sweeps + (a.offset if g else 0)
            o.run(betas[:n], update_slices_every=10)
            pos[g] = n
        for o in opts:
            o.sync()
        m0 = sum(o.counters()["moves"] for o in opts)
        t0 = time.perf_counter()
        for _s in range(a.steps):
            for g, o in enumerate(opts):
                o.run(betas[pos[g]:pos[g] + a.sweeps], update_slices_every=10)
                pos[g] += a.sweeps
        for o in opts:
            o.sync()
        dt = time.perf_counter() - t0
        m1 = sum(o.counters()["moves"] for o in opts)
        print(f"handles {G}: {(m1 - m0) / dt / 1e9:.3f}e9 move-evals/s, {dt / a.steps * 1e3:.2f} ms per {a.sweeps} sweeps", flush=True)
        for o in opts:
            o.close()


if __name__ == "__main__":
    main()
