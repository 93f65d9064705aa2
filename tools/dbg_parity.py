import sys; sys.path.insert(0,'.')
import numpy as np
from tests import helpers as H
from oracle import oracle as orc
from tnco_amd import core
prob=H.regular_problem(8,8,3)
betas=H.linear_betas(0,50,60)
seeds=H.replica_seeds(4,S=8)
links=prob.links(seeds)
o=H.make_oracle(orc,prob,links[0],seeds[0])
prev=None
for N in range(1,16):
    gpu=core.BatchedOptimizer(prob.leaf_masks,links[:1],seeds[:1],n_inds=prob.n_inds)
    gpu.run(betas[:N])
    o.update(2,betas[N-1])
    same=all(np.array_equal(a,b_) for a,b_ in zip(gpu.tree(0)[:3],o.tree()[:3]))
    print(N,"gpu pos",gpu.prng_state(0)[624],"moves",gpu.moves_per_replica()[0],"| oracle pos",o.prng_state()[624],"moves",o.counters()['moves'],"OK" if same else "DIFF")
    if not same:
        print("gpu ",*[x.tolist() for x in gpu.tree(0)[:3]]); print("orc ",*[x.tolist() for x in o.tree()[:3]])
        print("prev",*[x.tolist() for x in prev]); break
    prev=[x.copy() for x in o.tree()[:3]]
