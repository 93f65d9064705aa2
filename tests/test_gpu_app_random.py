"""Randomised tests of the plugin API on the GPU (hypothesis, fixed examples; TNCO_FUZZ=N for fresh
ones): random index-list networks with several connected components, hyper-indices, output legs
and single tensors go through Optimizer(method='sa').optimize(...), with and without pre-fusing,
with and without a width bound; every returned path is replayed symbolically
(as tests/test_contraction.py:57-352 of the reference replays min_ctree.path())."""
import math
import os
import warnings
from decimal import Decimal

import pytest
from hypothesis import given, settings, strategies as st

from tnco_amd import synthetic as syn
from tnco_amd.app import Optimizer, load_tn

pytestmark = pytest.mark.gpu

_N = int(os.environ.get("TNCO_FUZZ", "0"))
SET = dict(max_examples=_N or 10, derandomize=not _N, deadline=None)


def _spec(seed, sizes, k, lonely):
    """Index list [(dim, tensor names..., '*' for an output leg)] of a network with len(sizes)
    components (+ `lonely` tensors that share nothing)."""
    lines = []
    for c, n in enumerate(sizes):
        ts, _dims, out = syn.random_hyper_tn(n, int(1.8 * n) + 1, k=k, n_output=(seed + c) % 3, seed=seed + 17 * c)
        n_inds = 1 + max(i for xs in ts for i in xs)
        for i in range(n_inds):
            holders = [f"c{c}t{t}" for t in range(n) if i in ts[t]]
            if holders:
                lines.append((2, *holders, *(["*"] if i in out else [])))
    for j in range(lonely):
        lines.append((2, f"lonely{j}", "*"))
    return lines


def _replay(tn, path, slices=frozenset()):
    """Symbolic contraction along a linear path: (remaining tensors, flops with `slices` kept open)."""
    ts = [frozenset(x) for x in tn.ts_inds]
    count = {}
    for xs in ts:
        for i in xs:
            count[i] = count.get(i, 0) + 1
    for i in tn.output_inds:
        count[i] += 1
    cost = 0
    widths = []
    for x, y in path:
        x, y = sorted((x, y))
        b = ts.pop(y)
        a = ts.pop(x)
        cost += math.prod(tn.dims[i] for i in a | b | slices)
        new = set(a ^ b)
        for i in a & b:
            count[i] -= 1
            if count[i] > 1 or (count[i] == 1 and (i in tn.output_inds or any(i in t for t in ts))):
                new.add(i)
        widths.append(sum(math.log2(tn.dims[i]) for i in new - slices))
        ts.append(frozenset(new))
    return ts, cost, widths


@settings(**SET)
@given(seed=st.integers(0, 10**6), sizes=st.lists(st.integers(2, 14), min_size=1, max_size=3),
       k=st.sampled_from([2, 3]), lonely=st.sampled_from([0, 0, 1]), fuse=st.sampled_from([None, None, 3, 5]))
def test_random_networks_through_the_api(seed, sizes, k, lonely, fuse):
    spec = _spec(seed, sizes, k, lonely)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tn0 = load_tn(spec, fuse=None, decompose_hyper_inds=False)
        tn, res = Optimizer(method="sa", seed=seed).optimize(spec, betas=(0, 30), n_steps=40, n_runs=5, fuse=fuse)
    assert len(res) == 5 and [r.cost for r in res] == sorted(r.cost for r in res)
    for r in res:
        # every component's path, over all tensors of the (fused) network, costs what is reported
        assert len(r.disconnected_paths) == len(r.disconnected_costs)
        for path, cost in zip(r.disconnected_paths, r.disconnected_costs):
            _ts, flops, _w = _replay(tn, path)
            assert Decimal("%g" % flops) == cost
        assert r.cost == sum(r.disconnected_costs, Decimal(0))
        # the merged path contracts the whole (fused) network to its output legs
        final, _f, _w = _replay(tn, r.path)
        assert len(final) == 1 and final[0] == tn.output_inds
        # with pre-fusing: fuse path + path contract the ORIGINAL network
        if "fuse_path" in tn.tags:
            final, _f, _w = _replay(tn0, list(tn.tags["fuse_path"]) + [tuple(p) for p in r.path])
            assert len(final) == 1 and final[0] == tn0.output_inds


@settings(**SET)
@given(seed=st.integers(0, 10**6), sizes=st.lists(st.integers(4, 14), min_size=1, max_size=2),
       k=st.sampled_from([2, 3]), max_width=st.sampled_from([2, 3, 5]), update_slices=st.sampled_from([1, 10]))
def test_random_networks_with_a_width_bound(seed, sizes, k, max_width, update_slices):
    spec = _spec(seed, sizes, k, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tn, res = Optimizer(method="sa", max_width=max_width, seed=seed).optimize(
            spec, betas=(0, 30), n_steps=40, n_runs=4, update_slices=update_slices, fuse=None)
    assert len(res) == 4 and [r.cost for r in res] == sorted(r.cost for r in res)
    for r in res:
        assert r.slices == frozenset().union(*r.disconnected_slices)
        for path, cost, sl in zip(r.disconnected_paths, r.disconnected_costs, r.disconnected_slices):
            _ts, flops, widths = _replay(tn, path, frozenset(sl))
            assert Decimal("%g" % flops) == cost
            assert all(w <= max_width for w in widths)
        final, _f, _w = _replay(tn, r.path)
        assert len(final) == 1 and final[0] == tn.output_inds
