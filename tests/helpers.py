"""Shared test helpers: synthetic problems, oracle runs, comparisons."""
from __future__ import annotations

import random

import numpy as np

from tnco_amd import ctree as ct
from tnco_amd import synthetic as syn


class Problem:
    """A tensor network flattened for both the oracle and the GPU path."""

    def __init__(self, ts_inds, dims, output_inds=(), sparse_inds=(), n_inds=None):
        self.ts_inds = [list(x) for x in ts_inds]
        self.n = len(self.ts_inds)
        self.n_inds = (max((max(x) for x in self.ts_inds if x), default=-1) + 1) if n_inds is None else n_inds
        self.W = ct.n_words(self.n_inds)
        self.dims = dims
        self.leaf_masks = ct.pack_masks(self.ts_inds, self.n_inds)
        self.output_mask = ct.pack_masks([list(output_inds)], self.n_inds)[0]
        self.sparse_mask = ct.pack_masks([list(sparse_inds)], self.n_inds)[0] if len(sparse_inds) else None
        self.holders = [[] for _ in range(self.n_inds)]
        for t, xs in enumerate(self.ts_inds):
            for i in xs:
                self.holders[i].append(t)

    def tree(self, seed):
        con = ct.random_contraction(self.holders, self.n, seed=int(seed) & 0xFFFFFFFF)
        return ct.tree_from_contraction(con, self.n)

    def links(self, seeds):
        out = np.empty((len(seeds), 3, 2 * self.n - 1), np.int32)
        for k, s in enumerate(seeds):
            out[k, 0], out[k, 1], out[k, 2] = self.tree(s)
        return out

    def node_masks(self, left, right):
        return ct.derive_inds(left, right, self.leaf_masks, self.output_mask)


def regular_problem(n, graph_seed, degree=3):
    ts, d, out = syn.random_regular_tn(n, degree, graph_seed)
    return Problem(ts, d, out)


def replica_seeds(R, S=0):
    """seeds = Random(S).choices(range(2**32), k=R) -- tnco/app/infinite_memory/sa.py:237."""
    return random.Random(S).choices(range(2**32), k=R)


def linear_betas(b0, b1, n_steps):
    """more_itertools.numeric_range(b0, b1, (b1-b0)/n_steps): b0 + k*step (sa.py:155)."""
    step = (b1 - b0) / n_steps
    return np.array([b0 + k * step for k in range(n_steps)], np.float64)


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
