"""Shared test helpers: synthetic problems, oracle runs, comparisons."""
from __future__ import annotations

import numpy as np

from tnco_amd.synthetic import Problem, linear_betas, regular_problem, replica_seeds  # noqa: F401


def make_oracle(orc, prob: Problem, links_r, seed, **kw):
    l, r, p = links_r
    inds = prob.node_masks(l, r)
    return orc.Oracle(l, r, p, inds, n_inds=prob.n_inds, dims=prob.dims, sparse=prob.sparse_mask,
                      seed=int(seed) & 0xFFFFFFFF, **kw)


def assert_replica_equal(gpu, r, o, check_min=True):
    """Bit-exact comparison of one GPU replica with its oracle twin."""
    for which in ((False, True) if check_min else (False,)):
        gl, gr, gp, gm = gpu.tree(r, which_min=which)
        ol, orr, op, om = o.tree(which_min=which)
        assert np.array_equal(gl, ol), f"replica {r} left (min={which})"
        assert np.array_equal(gr, orr), f"replica {r} right (min={which})"
        assert np.array_equal(gp, op), f"replica {r} parent (min={which})"
        assert np.array_equal(gm, om), f"replica {r} masks (min={which})"
    cc, pc, hy = gpu.caches(r)
    occ, opc, ohy = o.caches()
    assert np.array_equal(cc.view(np.uint64), occ.view(np.uint64)), f"replica {r} ccost"
    assert np.array_equal(pc.view(np.uint64), opc.view(np.uint64)), f"replica {r} partial"
    assert np.array_equal(hy, ohy), f"replica {r} hyper"
    assert np.array_equal(gpu.prng_state(r), o.prng_state()), f"replica {r} prng"
