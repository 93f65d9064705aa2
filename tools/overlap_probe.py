#!/usr/bin/env python3
"""overlap_probe.py -- the launch tail of `sa_run_kernel` and the streams that remove it.

65536 replicas at 512 leaves are 1024 blocks for 768 resident ones: the last 256 blocks of every launch
run on a third of the chip.  A handle therefore splits every step over G streams (`TNCO_HIP_GROUPS`,
default 2 when the last round of blocks would be partial; csrc/host_ctx.h, tnco_hip_run): a block that
ends frees its slot for a block of the OTHER stream's pending launch, so the chip stays full from step to
step.  This tool times the bench's headline loop for G = 1, 2, 3, 4 (and checks that the best cost does
not depend on G: a replica's sweeps stay in order).

    python tools/overlap_probe.py [--groups 1,2,3,4] [--steps 20] [--sweeps 100]
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
    ap.add_argument("--groups", default="1,2,3,4")
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
    import os
    for G in [int(x) for x in a.groups.split(",")]:
        os.environ["TNCO_HIP_GROUPS"] = str(G)  # (read by tnco_hip_create)
        opt = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=2, device=0)
        for s in range(a.warmup):
            opt.run(betas[s * a.sweeps:(s + 1) * a.sweeps])
        opt.sync()
        m0 = opt.counters()["moves"]
        opt.kernel_times_ms(reset=True)
        t0 = time.perf_counter()
        for s in range(a.warmup, a.warmup + a.steps):
            opt.run(betas[s * a.sweeps:(s + 1) * a.sweeps])
        opt.sync()
        dt = time.perf_counter() - t0
        kt = opt.kernel_times_ms()
        m1 = opt.counters()["moves"]
        best = float(opt.costs()[1].min())
        print(f"groups {G}: {(m1 - m0) / dt / 1e9:.3f}e9 move-evals/s, {dt / a.steps * 1e3:.2f} ms per step (device "
              f"{kt['sa_run_kernel'][0] / a.steps:.2f}), best log10 {np.log10(best):.4f}", flush=True)
        opt.close()


if __name__ == "__main__":
    main()
