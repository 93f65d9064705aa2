"""Odd shapes through the LDS-resident kernels (csrc/sa_small.h) against the oracle: 2 ... 12 tensors sharing 3 ... 1000 indices
(1 ... 16 mask words: every K of sa_lds_kernel, index lists of up to hundreds of positions per leaf -> the table form),
hyper-indices and open indices at random, 3 ... 40 replicas, two launches.  60 cases; prints one line each and the failures.
Run on the GPU box: python tools/fuzz_shapes.py"""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests import helpers as H
from tnco_amd import core
from oracle import oracle as orc
orc.build()
rng = random.Random(5)
bad = 0
for case in range(60):
    n = rng.choice([2, 3, 4, 5, 7, 12])
    I = rng.choice([3, 70, 130, 200, 400, 700, 1000])
    ts = [[] for _ in range(n)]
    for i in range(I):
        k = rng.choice([2, 2, 2, 3, 4]) if n > 2 else 2
        for t in rng.sample(range(n), min(k, n)):
            ts[t].append(i)
    out = rng.sample(range(I), rng.choice([0, 0, 3]))
    for t in range(n):
        if not ts[t]: ts[t].append(rng.randrange(I))
    try:
        prob = H.Problem(ts, 2, out, n_inds=I)
        R = rng.choice([3, 17, 40])
        seeds = H.replica_seeds(R, S=case)
        links = prob.links(seeds)
        betas = H.linear_betas(0, 30, 25)
        with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, output_mask=prob.output_mask) as gpu:
            lg = gpu.launch_groups
            gpu.run(betas[:7]); gpu.run(betas[7:])
            tot, mn = gpu.costs()
            for r in range(0, R, 2):
                o = H.make_oracle(orc, prob, links[r], seeds[r])
                o.run(orc.PROB_MH, betas)
                H.assert_replica_equal(gpu, r, o)
                assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
            assert gpu.validate() == (0, -1)
        print(case, n, I, R, "lds" if lg == 0 else "hbm", "ok", flush=True)
    except ValueError as e:
        print(case, n, I, "skipped:", str(e)[:60])
    except Exception as e:
        bad += 1
        print(case, n, I, "FAILED", type(e).__name__, str(e)[:200], flush=True)
print("failures", bad)
