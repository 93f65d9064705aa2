"""Step-by-step comparison of the GPU path with the oracle (debugging aid):
    python tools/dbg_parity.py [n_leaves] [replicas] [sweeps]"""
import sys; sys.path.insert(0,'.')
import numpy as np
from tests import helpers as H
from oracle import oracle as orc
from tnco_amd import core
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = int(sys.argv[3]) if len(sys.argv) > 3 else 15
prob=H.regular_problem(n,8,3)
betas=H.linear_betas(0,50,60)
seeds=H.replica_seeds(R,S=8)
links=prob.links(seeds)
os_=[H.make_oracle(orc,prob,links[r],seeds[r]) for r in range(R)]
for N in range(1,S+1):
    gpu=core.BatchedOptimizer(prob.leaf_masks,links,seeds,n_inds=prob.n_inds)
    gpu.run(betas[:N])
    bad=[]
    for r,o in enumerate(os_):
        o.update(2,betas[N-1])
        same=all(np.array_equal(a,b_) for a,b_ in zip(gpu.tree(r)[:3],o.tree()[:3]))
        same_cost = gpu.costs()[0][r] == o.total_cost
        same_rng = np.array_equal(gpu.prng_state(r), o.prng_state())
        if not (same and same_cost and same_rng): bad.append((r,same,same_cost,same_rng,int(gpu.moves_per_replica()[r]),o.counters()['moves']))
    print(N, "bad:", bad[:6])
    if bad:
        r = bad[0][0]
        print("replica", r)
        print("gpu ", *[x.tolist() for x in gpu.tree(r)[:3]], gpu.prng_state(r)[624], gpu.costs()[0][r])
        print("orc ", *[x.tolist() for x in os_[r].tree()[:3]], os_[r].prng_state()[624], os_[r].total_cost)
        print("init", links[r].tolist())
        break
