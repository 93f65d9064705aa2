"""Property tests (hypothesis) of the host logic on either side of the path and of the oracle's
cost bookkeeping: path <-> tree conversions, fusing, connected components, and the oracle's caches
against a from-scratch recomputation in independent Python after random sweeps (the invariant the
reference's own test asserts every 10 sweeps, tests/test_utils.py:575-769)."""
import math
import random

import numpy as np
from hypothesis import given, settings, strategies as st

from tests import helpers as H
from tnco_amd import ctree as ct
from tnco_amd import synthetic as syn
from tnco_amd.app.infinite_memory.sa import merge_contraction_paths
from tnco_amd.app.tn import contract, fuse, get_connected_components

FAST = dict(max_examples=25, deadline=None)


def _random_linear_path(n, rng):
    path, m = [], n
    while m > 1:
        i, j = rng.sample(range(m), 2)
        path.append((i, j))
        m -= 1
    return path


@settings(**FAST)
@given(n=st.integers(2, 40), seed=st.integers(0, 2**31))
def test_linear_ssa_round_trip_and_tree(n, seed):
    rng = random.Random(seed)
    path = _random_linear_path(n, rng)
    ssa = ct.linear_to_ssa(path, n)
    assert [z for _x, _y, z in ssa] == list(range(n, 2 * n - 1))        # ids in creation order
    assert ct.ssa_to_linear(ssa, n) == [tuple(sorted(p)) for p in path]
    left, right, parent = ct.tree_from_contraction(ssa, n)
    N = 2 * n - 1
    assert parent[N - 1] == -1 and all(left[i] < 0 and right[i] < 0 for i in range(n))
    for p in range(n, N):
        assert parent[left[p]] == p and parent[right[p]] == p
    order = ct.traverse(left, right)
    assert sorted(order) == list(range(N)) and order[-1] == N - 1
    seen = set()
    for p in order:                                                     # children before parents
        if left[p] >= 0:
            assert left[p] in seen and right[p] in seen
        seen.add(p)
    again = ct.tree_from_contraction(ct.get_contraction(left, right), n)  # the same tree, renumbered
    leafsets = lambda l, r: {frozenset(_below(l, r, p)) for p in range(n, N)}  # noqa: E731
    assert leafsets(left, right) == leafsets(again[0], again[1])


def _below(left, right, p):
    out, stack = [], [p]
    while stack:
        x = stack.pop()
        if left[x] < 0:
            out.append(x)
        else:
            stack += [left[x], right[x]]
    return out


@settings(**FAST)
@given(sizes=st.lists(st.integers(1, 6), min_size=1, max_size=5), seed=st.integers(0, 2**31))
def test_merge_contraction_paths_contracts_everything(sizes, seed):
    """tnco/utils/tn.py:334-401: per-component paths (positions over ALL tensors) merge into one path
    that ends with a single tensor."""
    rng = random.Random(seed)
    n = sum(sizes)
    ids = list(range(n))
    rng.shuffle(ids)
    comps, lo = [], 0
    for s in sizes:
        comps.append(sorted(ids[lo:lo + s]))
        lo += s
    paths = []
    for comp in comps:
        pos, path, alive = list(range(n)), [], list(comp)
        while len(alive) > 1:
            a, b = rng.sample(alive, 2)
            ia, ib = sorted((pos.index(a), pos.index(b)))
            path.append((ia, ib))
            new = ("c", len(paths), len(path))
            pos.pop(ib)
            pos.pop(ia)
            pos.append(new)
            alive = [x for x in alive if x not in (a, b)] + [new]
        paths.append(path)
    merged = merge_contraction_paths(n, paths)
    m = n
    for x, y in merged:
        assert 0 <= x < y < m
        m -= 1
    assert m == 1


@settings(**FAST)
@given(seed=st.integers(0, 10**6), k=st.sampled_from([2, 3]), width=st.sampled_from([0.0, 2.0, 4.0, 6.0, math.inf]))
def test_fuse_is_a_valid_width_bounded_path(seed, k, width):
    ts, dims, out = syn.random_hyper_tn(10, 14, k=k, n_output=2, seed=seed, dims_choices=(2, 4))
    used = {x for xs in ts for x in xs}
    dims = {x: d for x, d in enumerate(dims) if x in used}
    path, fused = fuse(ts, dims, width, output_inds=out, seed=seed, return_fused_inds=True)
    assert all(sum(math.log2(dims[x]) for x in zs) <= width for zs in fused)
    new_ts, new_out = contract(path, ts, out, dims=dims)
    assert len(new_ts) == len(ts) - len(path) and new_out == frozenset(out)
    assert len(get_connected_components(new_ts)) == len(get_connected_components(ts))
    if width == math.inf:
        assert len(new_ts) == len(get_connected_components(ts))
    if width == 0.0:
        assert path == []


@settings(max_examples=8, deadline=None)
@given(seed=st.integers(0, 10**6), kind=st.sampled_from(["plain", "dims3", "vector", "sparse"]))
def test_oracle_caches_equal_brute_force(oracle_lib, seed, kind):
    """After random sweeps: contraction costs, partial costs (association order of
    infinite_memory/utils.hpp:54), hyper legs and leg masks of the oracle equal a recomputation
    from the tree alone; min <= current; the tree is valid."""
    ts, dims, out = syn.random_hyper_tn(14, 30, k=3, n_output=2, seed=seed, dims_choices=(2, 3, 4))
    sparse = []
    if kind == "plain":
        d = 2
    elif kind == "dims3":
        d = 3
    elif kind == "vector":
        d = np.array(dims, np.uint64)
    else:
        d, sparse = 2, [0, 3, 7, 12, 20]
    prob = H.Problem(ts, d, out, sparse_inds=sparse)
    s = seed & 0xFFFFFFFF
    left, right, parent = prob.tree(s)
    kw = dict(n_projs=4) if sparse else {}
    o = H.make_oracle(oracle_lib, prob, (left, right, parent), s, **kw)
    o.run(oracle_lib.PROB_MH, H.linear_betas(0, 20, 25))
    assert o.is_valid() == 0 and o.min_total_cost <= o.total_cost
    l, r, p, masks = o.tree()
    cc, pc, hy = o.caches()
    n, N = prob.n, 2 * prob.n - 1
    assert np.array_equal(masks[:n], prob.leaf_masks)
    dimlist = [int(x) for x in (d if np.ndim(d) else [d] * prob.n_inds)]
    sp = set(sparse)

    def cost(legs):
        a = b = 1.0
        for x in sorted(legs):
            if x in sp:
                b *= dimlist[x]
            else:
                a *= dimlist[x]
        return a * (min(b, 4.0) if sp else b)

    part = [0.0] * N
    for q in ct.traverse(l, r):
        if l[q] < 0:
            assert cc[q] == 0.0 and pc[q] == 0.0
            continue
        la, lb, lq = (set(ct.unpack_mask(masks[x])) for x in (l[q], r[q], q))
        assert cc[q] == cost(la | lb)
        part[q] = (cc[q] + part[l[q]]) + part[r[q]]
        assert set(ct.unpack_mask(hy[q])) == lq & la & lb
        assert (la ^ lb) <= lq <= (la | lb)
    # the running partial sums are updated incrementally with another association order
    # (optimizer.hpp:185-188): equal up to rounding, as the reference's is_valid(atol) allows
    for q in range(n, N):
        assert math.isclose(pc[q], part[q], rel_tol=1e-12)
