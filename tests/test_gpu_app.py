"""GPU tests of the plugin API: tnco_amd.app.Optimizer(method='sa').optimize(...)
(the call shape of /root/reference/README.md:93-106 and tests/test_app.py:117-329)."""
import json
import math
import random
from decimal import Decimal

import numpy as np
import pytest

from tests import helpers as H
from tnco_amd import ctree as ct
from tnco_amd.app import Optimizer, load_tn
from tnco_amd.app.infinite_memory.sa import expand_betas

pytestmark = pytest.mark.gpu


def _replay(tn, path):
    """Contract symbolically along a linear path: returns (final legs, total cost)
    (tests/test_contraction.py:57-181 replays min_ctree.path() the same way)."""
    ts = [frozenset(x) for x in tn.ts_inds]
    count = {}
    for xs in ts:
        for i in xs:
            count[i] = count.get(i, 0) + 1
    for i in tn.output_inds:
        count[i] += 1
    cost = 0
    for x, y in path:
        x, y = sorted((x, y))
        b = ts.pop(y)
        a = ts.pop(x)
        cost += math.prod(tn.dims[i] for i in a | b)
        out = set(a ^ b)
        for i in a & b:
            count[i] -= 1
            if count[i] > 1 or (count[i] == 1 and (i in tn.output_inds or any(i in t for t in ts))):
                out.add(i)
        ts.append(frozenset(out))
    return ts, cost


def test_readme_chain_config1():
    """BASELINE config 1: 3-index chain of the README, n_steps=100, n_runs=8 (un-fused)."""
    opt = Optimizer(method="sa", seed=0)
    tn, res = opt.optimize("2 a b\n2 b c\n2 c d", betas=(0, 100), n_steps=100, n_runs=8, fuse=None)
    assert len(res) == 8 and [r.cost for r in res] == sorted(r.cost for r in res)
    for r in res:
        final, cost = _replay(tn, r.path)
        assert len(final) == 1 and final[0] == tn.output_inds
        assert Decimal("%g" % cost) == r.cost
        assert len(r.path) == 3 and len(r.disconnected_paths) == 1
        assert [tuple(sorted(p)) for p in r.disconnected_paths[0]] == [tuple(p) for p in r.path]
    # the optimum of the open 4-chain: sweep from one end, 4 + 4 + 2 flops
    assert res[0].cost == Decimal(10)


@pytest.mark.parametrize("initial_trees", ["greedy", "kruskal"])
def test_results_match_oracle_run_by_run(oracle_lib, initial_trees):
    ts, dims, out = __import__("tnco_amd.synthetic", fromlist=["x"]).random_regular_tn(24, 3, 5)
    spec = [(2, *[f"t{t}" for t in range(len(ts)) if k in ts[t]]) for k in range(36)]
    opt = Optimizer(method="sa", seed=11)
    n_runs, n_steps = 12, 150
    tn, res = opt.optimize(spec, betas=(0, 50), n_steps=n_steps, n_runs=n_runs, fuse=None,
                           initial_trees=initial_trees)
    seeds = random.Random(11).choices(range(2**32), k=n_runs)
    betas = expand_betas((0, 50), n_steps)
    # bit position of an index = order of first appearance over the tensors (tnco/ctree.py:232,
    # `_inds_order`); the initial-tree generator walks indices by position
    imap = {x: k for k, x in enumerate(dict.fromkeys(i for xs in tn.ts_inds for i in xs))}
    prob = H.Problem([[imap[i] for i in xs] for xs in tn.ts_inds], 2)
    want = []
    for s in seeds:
        if initial_trees == "greedy":  # the reference's recipe (tnco/utils/tn.py:189-230), Python spec
            l, r, p = ct.tree_from_contraction(ct.greedy_contraction(prob.ts_inds, (), s), prob.n)
        else:
            l, r, p = prob.tree(s)
        o = H.make_oracle(oracle_lib, prob, (l, r, p), s)
        o.run(oracle_lib.PROB_MH, betas)
        ml, mr, _mp, _ = o.tree(which_min=True)
        want.append((Decimal("%g" % o.min_total_cost), ct.ssa_to_linear(ct.get_contraction(ml, mr), len(ts))))
    want.sort(key=lambda t: t[0])
    assert [r.cost for r in res] == [w[0] for w in want]
    got_paths = sorted(tuple(map(tuple, r.disconnected_paths[0])) for r in res)
    assert got_paths == sorted(tuple(map(tuple, w[1])) for w in want)
    for r in res:  # one component: the merged path is the component's path with sorted pairs
        assert [tuple(p) for p in r.path] == [tuple(sorted(p)) for p in r.disconnected_paths[0]]
    assert float(res[0].cost) <= float(tn.tags["best_raw_cost"]) * (1 + 1e-5)


def test_full_head_of_1024_runs_matches_the_oracle(oracle_lib):
    """`optimize(n_runs=1024)` returns all 1024 runs (top_k defaults to min(n_runs, 1024)): costs and
    paths of the whole list, assembled from the device-side extraction, against 1024 oracle runs."""
    ts, dims, out = __import__("tnco_amd.synthetic", fromlist=["x"]).random_regular_tn(14, 3, 2)
    spec = [(2, *[f"t{t}" for t in range(len(ts)) if k in ts[t]]) for k in range(21)]
    n_runs, n_steps = 1024, 40
    tn, res = Optimizer(method="sa", seed=5).optimize(spec, betas=(0, 20), n_steps=n_steps, n_runs=n_runs, fuse=None)
    assert len(res) == n_runs
    seeds = random.Random(5).choices(range(2**32), k=n_runs)
    betas = expand_betas((0, 20), n_steps)
    imap = {x: k for k, x in enumerate(dict.fromkeys(i for xs in tn.ts_inds for i in xs))}
    prob = H.Problem([[imap[i] for i in xs] for xs in tn.ts_inds], 2)
    want = []
    for gid, s in enumerate(seeds):
        l, r, p = ct.tree_from_contraction(ct.greedy_contraction(prob.ts_inds, (), s), prob.n)
        o = H.make_oracle(oracle_lib, prob, (l, r, p), s)
        o.run(oracle_lib.PROB_MH, betas)
        ml, mr, _mp, _ = o.tree(which_min=True)
        want.append((Decimal("%g" % o.min_total_cost), gid, ct.ssa_to_linear(ct.get_contraction(ml, mr), len(ts))))
    want.sort(key=lambda t: (t[0], t[1]))  # sorted(results) is stable over the run order (sa.py:257)
    assert [r.cost for r in res] == [w[0] for w in want]
    assert [[tuple(p) for p in r.disconnected_paths[0]] for r in res] == [[tuple(p) for p in w[2]] for w in want]


def test_disconnected_components_and_json():
    spec = "2 a b\n2 b c\n3 x y\n3 y z\n3 z x\n2 lonely"
    opt = Optimizer(method="sa", seed=3, output_format="json")
    out = json.loads(opt.optimize(spec, betas=(0, 10), n_steps=30, n_runs=4, fuse=None))
    assert len(out["res"]) == 4 and len(out["tn"]["tensors"]) == 7
    tn = load_tn(spec, fuse=None)
    for r in out["res"]:
        assert len(r["path"]) == 6  # 7 tensors -> 6 contractions after autocomplete
        assert len(r["disconnected_paths"]) == 3
        final, _ = _replay(tn, [tuple(p) for p in r["path"]])
        assert len(final) == 1


def test_finite_width_through_the_api():
    """tests/test_app.py:117-329 + tests/test_contraction.py:184-352 of the reference: results carry
    `slices`; every intermediate tensor fits `max_width` once the sliced indices are removed; the
    cost is the sliced contraction's flops."""
    ts, dims, out = __import__("tnco_amd.synthetic", fromlist=["x"]).random_regular_tn(40, 3, 9)
    spec = [(2, *[f"t{t}" for t in range(40) if k in ts[t]]) for k in range(60)]
    max_width = 5
    tn, res = Optimizer(method="sa", max_width=max_width, seed=4).optimize(
        spec, betas=(0, 80), n_steps=200, n_runs=6, update_slices=10, fuse=None)
    assert len(res) == 6 and [r.cost for r in res] == sorted(r.cost for r in res)
    for r in res:
        assert r.slices == frozenset().union(*r.disconnected_slices) and len(r.slices) > 0
        legs = [frozenset(x) for x in tn.ts_inds]
        cost = 0
        for x, y in r.path:
            x, y = sorted((x, y))
            b = legs.pop(y)
            a = legs.pop(x)
            cost += 2 ** len(a | b | r.slices)
            new = a ^ b
            assert len(new - r.slices) <= max_width
            legs.append(new)
        assert Decimal("%g" % cost) == r.cost


def test_readme_chain_default_fuse():
    """README.md:93-106 with the default `fuse=4`: every index of the 4-tensor chain is pre-contracted
    (all intermediates have width <= 2), one tensor without indices is left and the SA loop is skipped
    (tnco/app/infinite_memory/sa.py:179-183): cost 0, empty path, tn.tags['fuse_path'] holds the fusing."""
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tn, res = Optimizer(method="sa", seed=0).optimize("2 a b\n2 b c\n2 c d", betas=(0, 100), n_steps=100, n_runs=8)
    assert any("hyper-indices" in str(x.message) for x in w)
    assert len(tn.tensors) == 1 and tn.tensors[0].inds == () and len(tn.tags["fuse_path"]) == 3
    assert len(res) == 8
    for r in res:
        assert r.cost == 0 and list(r.path) == [] and list(r.disconnected_paths) == [[]]


def test_fused_network_runs_on_the_gpu():
    """fuse=4 on a 3-regular network merges some tensors; the optimizer runs on the fused network
    and fuse_path + path contract the original network to its outputs."""
    ts, dims, out = __import__("tnco_amd.synthetic", fromlist=["x"]).random_regular_tn(40, 3, 3)
    spec = [(2, *[f"t{t}" for t in range(40) if k in ts[t]]) for k in range(60)]
    tn0 = load_tn(spec, fuse=None, decompose_hyper_inds=False)
    tn, res = Optimizer(method="sa", seed=2).optimize(spec, betas=(0, 50), n_steps=100, n_runs=4, fuse=4,
                                                      decompose_hyper_inds=False)
    assert 1 < len(tn.tensors) < 40
    final, _ = _replay(tn0, list(tn.tags["fuse_path"]) + [tuple(p) for p in res[0].path])
    assert len(final) == 1 and final[0] == tn0.output_inds
    final, cost = _replay(tn, res[0].path)
    assert Decimal("%g" % cost) == res[0].cost


def test_timeout_and_top_k():
    ts, dims, out = __import__("tnco_amd.synthetic", fromlist=["x"]).random_regular_tn(64, 3, 7)
    spec = [(2, *[f"t{t}" for t in range(64) if k in ts[t]]) for k in range(96)]
    # (200 000 sweeps: ten times the 0.2 s -- round 5's LDS-resident kernel runs 20 000 of them in 0.17 s)
    tn, res = Optimizer(method="sa", seed=1).optimize(spec, betas=(0, 100), n_steps=200000, n_runs=64,
                                                      timeout=0.2, top_k=5, sweeps_per_launch=50, fuse=None)
    assert len(res) == 5 and tn.tags["timed_out"] and tn.tags["n_runs"] == 64


def test_progress_series_is_non_increasing_and_ends_at_the_best_result(capsys):
    """`verbose` / `progress`: best min_total_cost so far per launch chunk (replaces the reference's `status` /
    `log2_total_cost` buffers, tnco/parallel.py:229-317, tnco/app/infinite_memory/sa.py:208-209)."""
    ts, dims, out = __import__("tnco_amd.synthetic", fromlist=["x"]).random_regular_tn(64, 3, 7)
    spec = [(2, *[f"t{t}" for t in range(64) if k in ts[t]]) for k in range(96)]
    seen = []
    tn, res = Optimizer(method="sa", seed=3, verbose=True).optimize(
        spec, betas=(0, 100), n_steps=600, n_runs=256, sweeps_per_launch=25, fuse=None,
        progress=lambda comp, done, total, best: seen.append((comp, done, total, best)))
    series = tn.tags["progress"]
    assert len(series) == len(seen) >= 10 and series[-1]["sweeps"] == series[-1]["of"] == 600
    costs = [p["best_cost"] for p in series]
    assert all(a >= b for a, b in zip(costs, costs[1:])) and costs[0] > costs[-1]
    assert [s[3] for s in seen] == costs
    assert Decimal("%g" % costs[-1]) == res[0].cost and costs[-1] == tn.tags["best_raw_cost"]
    err = capsys.readouterr().err
    assert err.count("log2(min_total_cost)") == len(series)
    # the finite-width driver reports the same way
    tn, res = Optimizer(method="sa", seed=3, max_width=8).optimize(
        spec, betas=(0, 50), n_steps=100, n_runs=64, sweeps_per_launch=20, fuse=None,
        progress=lambda *a: seen.append(a))
    costs = [p["best_cost"] for p in tn.tags["progress"]]
    assert len(costs) == 5 and all(a >= b for a, b in zip(costs, costs[1:]))
    assert Decimal("%g" % costs[-1]) == res[0].cost
