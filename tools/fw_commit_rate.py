"""How often does a re-slice keep its new slices (greedy/optimizer.hpp:371-374)?  Config-5 topology;
the moves never change the slices (max_number_new_slices = 0), so a changed mask = a kept re-slice."""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import core, synthetic  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = synthetic.sycamore_problem(20)
seeds = synthetic.replica_seeds(R)
links = core.random_trees(p.ts_inds, p.n_inds, seeds)
opt = core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=40)
betas = synthetic.linear_betas(0, 100, 1200)
ids = np.arange(R)
prev = opt.slices_many(ids)[0]
for c in range(0, 1200, 10):
    opt.run(betas[c:c + 10], update_slices_every=10)
    cur = opt.slices_many(ids)[0]
    if c % 100 == 0:
        ch = np.any(cur != prev, axis=1).mean()
        ns = np.unpackbits(cur.view(np.uint8), axis=1).sum(axis=1).mean()
        print(f"sweep {c:5d}: re-slices kept {ch:.3f}   mean number of slices {ns:.1f}   moves/sweep so far "
              f"{opt.counters()['moves'] / R / (c + 10):.1f}")
    prev = cur
