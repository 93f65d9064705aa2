"""CPU tests of the host logic: tree flattening, paths, initial trees, schedule, loader, sharding."""
import random

import numpy as np
import pytest

from tests import helpers as H
from tnco_amd import ctree as ct
from tnco_amd import parallel
from tnco_amd import synthetic as syn
from tnco_amd.app import Optimizer, load_tn
from tnco_amd.app.infinite_memory.sa import expand_betas, merge_contraction_paths


def _random_linear_path(n, rng):
    path, m = [], n
    while m > 1:
        i, j = rng.sample(range(m), 2)
        path.append((i, j))
        m -= 1
    return path


def _clusters(tree):
    """Set of leaf sets of the internal nodes: the tree up to renumbering of intermediates."""
    below = {}
    for p in ct.traverse(tree.left, tree.right):
        below[p] = frozenset([p]) if tree.left[p] < 0 else below[tree.left[p]] | below[tree.right[p]]
    return {v for p, v in below.items() if tree.left[p] >= 0}


def test_contraction_tree_matches_derive_and_roundtrips():
    """ctree.py:108-251 flattening (hyper-count bookkeeping) == set rule; path() round trip
    (tests/test_utils.py:351-572)."""
    rng = random.Random(1)
    for seed in range(8):
        ts, dims, out = syn.random_hyper_tn(12, 26, k=3, n_output=3, seed=seed, dims_choices=(2, 3))
        names = [[f"i{p}" for p in xs] for xs in ts]
        path = _random_linear_path(len(ts), rng)
        tree = ct.ContractionTree(path, names, {f"i{p}": d for p, d in enumerate(dims)},
                                  output_inds=[f"i{p}" for p in out])
        n = tree.n_leaves
        assert len(tree) == 2 * n - 1 and tree.parent[-1] == -1
        assert all(tree.left[i] < 0 for i in range(n))
        derived = ct.derive_inds(tree.left, tree.right, tree.masks[:n], tree.output_mask)
        assert np.array_equal(derived, tree.masks)
        # path() reproduces a path whose tree is the same tree
        again = ct.ContractionTree(tree.path(), names, {f"i{p}": d for p, d in enumerate(dims)},
                                   output_inds=[f"i{p}" for p in out])
        assert _clusters(again) == _clusters(tree)  # same tree up to the numbering of intermediates
        # linear -> SSA -> linear keeps every step up to the order inside a pair (pairs are sorted
        # on the way in, ctree.py:117)
        assert ct.ssa_to_linear(ct.linear_to_ssa(path, len(ts)), len(ts)) == [tuple(sorted(p)) for p in path]


def test_readme_example_tree():
    """README.md:93-106 input '2 a b / 2 b c / 2 c d': 4 tensors in a chain."""
    tn = load_tn("2 a b\n2 b c\n2 c d", fuse=None, decompose_hyper_inds=False)
    assert len(tn) == 4 and tn.output_inds == frozenset() and set(tn.dims.values()) == {2}
    tree = ct.ContractionTree([(0, 1), (0, 1), (0, 1)], tn.ts_inds, 2, check_shared_inds=True)
    assert tree.n_leaves == 4 and tree.max_width() == 2.0
    with pytest.raises(ValueError):
        ct.ContractionTree([(0, 3), (0, 1), (0, 1)], tn.ts_inds, 2, check_shared_inds=True)


def test_random_contraction_is_valid_and_native_matches():
    from tnco_amd import core
    for n, seed in [(2, 0), (8, 1), (64, 2), (200, 3)]:
        deg = 3 if n > 4 else 1
        prob = H.regular_problem(n, graph_seed=seed, degree=deg) if n > 2 else H.Problem([[0], [0]], 2)
        seeds = [0, 1, 2**32 - 1, 123456789]
        links = core.random_trees(prob.ts_inds, prob.n_inds, seeds)
        for k, s in enumerate(seeds):
            l, r, p = prob.tree(s)
            assert np.array_equal(links[k], np.stack([l, r, p]))
            # every contraction shares an index (check_shared_inds, sa.py:186-190)
            ct.derive_inds(l, r, prob.leaf_masks, prob.output_mask, check_shared_inds=True)
    with pytest.raises(ValueError):
        core.random_trees([[0], [0], [1], [1]], 2, [1])  # two components


def _greedy_case(ts, n_inds, out_keep, seeds):
    from tnco_amd import core
    om = ct.pack_masks([list(out_keep)], n_inds)[0]
    links = core.greedy_trees(ts, n_inds, seeds, output_mask=om)
    for k, s in enumerate(seeds):
        con = ct.greedy_contraction(ts, out_keep, int(s))
        l, r, p = ct.tree_from_contraction(con, len(ts))
        assert np.array_equal(links[k], np.stack([l, r, p])), (len(ts), s)
    return links


def test_greedy_initial_trees_native_matches_the_python_spec():
    """tnco/utils/tn.py:189-230 restated twice (Python with heapq / frozensets / random.Random, C++
    batched): same trees.  Random(seed).shuffle is CPython's own in the spec, so the native MT19937 +
    _randbelow is checked against this image's interpreter."""
    seeds = [0, 1, 42, 2**32 - 1, 123456789, 77]
    for n, gs in [(3, 1), (8, 1), (64, 7), (200, 3)]:
        prob = H.regular_problem(n, graph_seed=gs, degree=3 if n > 4 else 2)
        links = _greedy_case(prob.ts_inds, prob.n_inds, (), seeds)
        for k in range(len(seeds)):  # every contraction shares an index (check_shared_inds, sa.py:186-190)
            ct.derive_inds(links[k, 0], links[k, 1], prob.leaf_masks, prob.output_mask, check_shared_inds=True)
    # hyper-indices (held by 3 tensors), output legs held by one tensor, tensors with EQUAL index
    # sets (eager Hadamard products), an index common to all tensors (joins the output)
    from tnco_amd import synthetic as syn
    for seed in range(6):
        ts, _d, out = syn.random_hyper_tn(20, 37, k=3, n_output=4, seed=seed)
        n_inds = 1 + max(i for xs in ts for i in xs)
        cnt = [sum(i in xs for xs in ts) for i in range(n_inds)]
        _greedy_case(ts, n_inds, [i for i in out if cnt[i] <= 1], seeds[:3])
    ts = [[0, 1], [0, 1], [1, 2], [2, 3], [1, 2], [3, 0]]
    _greedy_case(ts, 4, (), seeds)
    ts = [[0, 1, 9], [1, 2, 9], [2, 3, 9], [3, 4, 9], [4, 0, 9]]
    _greedy_case(ts, 10, (), seeds)
    # the docstring example of the reference (tn.py:147-150): i-j, j-k, k-l with outputs i, l, seed 42
    con = ct.greedy_contraction([[0, 1], [1, 2], [2, 3]], [0, 3], 42)
    assert ct.ssa_to_linear(con, 3) == [(0, 1), (0, 1)]


def _small_hyper_networks(count, seed0=0):
    """Small random hyper networks in which equal index sets reappear DURING the greedy contraction (an
    in-loop Hadamard product with a live tensor): indices on 3-4 tensors, duplicated tensors."""
    from tnco_amd import synthetic as syn
    out = []
    for s in range(seed0, seed0 + count):
        rng = np.random.RandomState(1000 + s)
        n = int(rng.randint(5, 13))
        ts, _d, outs = syn.random_hyper_tn(n, int(n * rng.uniform(1.0, 1.8)), k=int(rng.randint(3, 5)),
                                           n_output=int(rng.randint(0, 3)), seed=s)
        for _ in range(int(rng.randint(0, 3))):  # tensors with the index set of another one
            ts.append(list(ts[int(rng.randint(0, len(ts)))]))
        n_inds = 1 + max(i for xs in ts for i in xs)
        cnt = [sum(i in xs for xs in ts) for i in range(n_inds)]
        out.append((ts, n_inds, [i for i in outs if cnt[i] <= 1]))
    return out


def test_greedy_initial_trees_fuzz_on_hyper_networks():
    """Round-2 advisor finding: when a contraction's result equals a LIVE index set, the two tensors
    that just left must also leave the neighbour row of that set (opt_einsum drops them from
    dim_to_keys before it looks for new candidates) -- 62 of 1727 small hyper networks differed from
    the spec before the fix, none of the handful the older test draws."""
    seeds = [0, 1, 2, 3, 5, 8, 13, 21]
    for ts, n_inds, out_keep in _small_hyper_networks(300):
        _greedy_case(ts, n_inds, out_keep, seeds)


def test_greedy_initial_trees_share_one_generator_over_components():
    """tn.py:163,192: one Random(seed) for all components of a run; the native call continues from
    the number of outputs consumed so far."""
    from random import Random
    from tnco_amd import core
    a, b = H.regular_problem(10, graph_seed=2), H.regular_problem(16, graph_seed=4)
    seeds = [3, 99, 2**31 + 5]
    draws = np.zeros(len(seeds), np.uint64)
    la = core.greedy_trees(a.ts_inds, a.n_inds, seeds, draws=draws)
    assert np.all(draws >= 9)
    lb = core.greedy_trees(b.ts_inds, b.n_inds, seeds, draws=draws)
    for k, s in enumerate(seeds):
        rng = Random(s)
        for prob, links in ((a, la), (b, lb)):
            con = ct.greedy_contraction(prob.ts_inds, (), s, rng=rng)
            assert np.array_equal(links[k], np.stack(ct.tree_from_contraction(con, prob.n)))


def test_native_linear_paths_match_path_spec():
    """tnco/ctree.py:350-388 (`path()`): the native batch (Fenwick ranks) against the list-based spec."""
    from tnco_amd import core
    rng = np.random.RandomState(5)
    for n, gs in [(2, 0), (10, 1), (40, 2)]:
        prob = H.regular_problem(n, graph_seed=gs, degree=3 if n > 4 else 1) if n > 2 else H.Problem([[0], [0]], 2)
        n_tensors = n + 7
        tensors_pos = np.sort(rng.choice(n_tensors, n, replace=False)).astype(np.int32)
        cons = []
        for s in range(5):
            l, r, _p = prob.tree(s)
            cons.append(ct.get_contraction(l, r))
        got = core.linear_paths(np.array(cons, np.int32).reshape(5, n - 1, 3), tensors_pos, n_tensors)
        for k, con in enumerate(cons):
            shift = n_tensors - n
            resc = [tuple(int(tensors_pos[p]) if p < n else p + shift for p in xs) for xs in con]
            assert got[k].tolist() == [list(p) for p in ct.ssa_to_linear(resc, n_tensors)]
    with pytest.raises(ValueError):
        core.linear_paths(np.zeros((1, 3, 3), np.int32), np.arange(4, dtype=np.int32), 4)


def test_merge_contraction_paths_docstring_example():
    assert merge_contraction_paths(4, [[(0, 1)], [(2, 3)]]) == [(0, 1), (0, 1), (0, 1)]
    assert merge_contraction_paths(3, [[], [], []]) == [(0, 1), (0, 1)]


def test_expand_betas():
    b = expand_betas((0.0, 100.0), 100)
    assert len(b) == 100 and b[0] == 0.0 and b[1] == 1.0 and b[-1] == 99.0
    assert list(expand_betas((10.0, 0.0), 4)) == [10.0, 7.5, 5.0, 2.5]
    assert list(expand_betas([1, 2, 3], None)) == [1.0, 2.0, 3.0]
    assert list(expand_betas([1, 2, 3], 2)) == [1.0, 2.0]
    for bad in (0, -1, 1.5):
        with pytest.raises(ValueError, match="'n_steps' must be a positive number."):
            expand_betas((0, 1), bad)
    with pytest.raises(ValueError, match="must be provided"):
        expand_betas((0, 1), None)
    with pytest.raises(ValueError, match="beta_ini != beta_end"):
        expand_betas((1, 1), 5)


def test_load_tn_formats_and_tokens():
    with pytest.warns(UserWarning, match="sparse indices"):  # app.py:330-336: no fusing with sparse indices
        tn = load_tn([(2, "a", "b"), (3, "b", "c", "*"), (2, "c", "/")])
    assert len(tn) == 3 and tn.output_inds == {1} and tn.sparse_inds == {2} and tn.dims == {0: 2, 1: 3, 2: 2}
    assert load_tn(tn, fuse=None, decompose_hyper_inds=False) is tn
    with pytest.warns(UserWarning, match="Cannot decompose hyper-indices"):
        assert len(load_tn("2 a b", fuse=4)) == 1
    with pytest.raises(TypeError):
        load_tn("hello world")
    with pytest.raises(TypeError):
        load_tn(3.5)


def test_factory_dispatch_and_validation():
    opt = Optimizer(method="sa", seed=1)
    assert type(opt).__module__.endswith("app.infinite_memory.sa")
    assert type(Optimizer(method="sa", max_width=20)).__module__.endswith("app.finite_width.sa")
    assert type(Optimizer(method="sa", max_width=float("inf"))).__module__.endswith("app.infinite_memory.sa")
    with pytest.raises(ModuleNotFoundError):
        Optimizer(method="nope")
    with pytest.raises(ValueError):
        Optimizer(output_format="xml")
    with pytest.raises(ValueError, match="'n_steps' must be a positive number."):
        opt.optimize("2 a b\n2 b c", betas=(0, 1), n_steps=0)
    with pytest.raises(ValueError, match="'update_slices' must be a positive number."):
        Optimizer(method="sa", max_width=4).optimize("2 a b\n2 b c", betas=(0, 1), n_steps=3, update_slices=0)


def test_shard_bounds_partition():
    for n_runs in (1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n_runs, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n_runs
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def test_seed_list_is_the_reference_rule():
    """seeds = Random(seed).choices(range(2**32), k=n_runs) (sa.py:237)."""
    assert H.replica_seeds(3, S=0) == random.Random(0).choices(range(2**32), k=3)


def test_dump_results_formats_and_files(tmp_path):
    """dump_results (tnco/app/app.py:573-712): raw tuple, JSON string, pickle / JSON files with
    'auto' compression from the suffix, refusal to overwrite."""
    import bz2
    import gzip
    import json
    import pickle
    from decimal import Decimal
    from tnco_amd.app.app import dump_results
    from tnco_amd.app.infinite_memory.sa import ContractionResults
    tn = load_tn("2 a b\n2 b c", fuse=None, decompose_hyper_inds=False)
    res = [ContractionResults(cost=Decimal("8"), runtime_s=0.5, path=[(0, 1), (0, 1)],
                              disconnected_costs=[Decimal("8")], disconnected_paths=[[(0, 1), (0, 1)]])]
    assert dump_results(tn, res) == (tn, res)
    js = json.loads(dump_results(tn, res, output_format="json"))
    assert len(js["tn"]["tensors"]) == 3 and js["res"][0]["path"] == [[0, 1], [0, 1]]
    assert js["res"][0]["disconnected_paths"] == [[[0, 1], [0, 1]]]
    for name, opener in (("out.pkl", open), ("out.gzip", gzip.open), ("out.bz2", bz2.open)):
        f = tmp_path / name
        assert dump_results(tn, res, output_filename=f) is None
        with opener(f, "rb") as fh:
            t2, r2 = pickle.load(fh)
        assert t2.ts_inds == tn.ts_inds and r2[0].cost == Decimal("8") and r2[0].path == [(0, 1), (0, 1)]
        with pytest.raises(FileExistsError):
            dump_results(tn, res, output_filename=f)
        assert dump_results(tn, res, output_filename=f, overwrite_output_file=True) is None
    f = tmp_path / "out.json.gzip"
    dump_results(tn, res, output_format="json", output_filename=f)
    with gzip.open(f, "rb") as fh:
        assert json.loads(fh.read().decode())["res"][0]["runtime_s"] == 0.5
    f = tmp_path / "plain.json"
    dump_results(tn, res, output_format="json", output_filename=f, output_compression="none")
    assert json.loads(f.read_text())["res"][0]["path"] == [[0, 1], [0, 1]]
    with pytest.raises(ValueError):
        dump_results(tn, res, output_compression="zip")
    with pytest.raises(TypeError):
        dump_results(tn, res, nope=1)
    assert res[0] < ContractionResults(cost=Decimal("9"), runtime_s=0, path=[], disconnected_costs=[], disconnected_paths=[])


def test_greedy_cost_key_orders_like_the_exact_cost():
    """csrc/greedy_key.h: the 36-bit key the device generator of the initial trees sorts opt_einsum's
    candidates by must order 2^a - 2^b - 2^c exactly like the integers do (ties included): every
    (a, b, c) up to 22, and exponents far apart / at the upper limit."""
    from tnco_amd import _lib
    L = _lib.load()
    def check(triples):
        vals = sorted(((1 << a) - (1 << b) - (1 << c), L.tnco_hip_diag_greedy_cost_key(a, b, c)) for a, b, c in triples)
        for (v0, k0), (v1, k1) in zip(vals, vals[1:]):
            assert (k0 < k1) if v0 < v1 else (k0 == k1), (v0, v1, k0, k1)
    rng = range(23)
    check([(a, b, c) for a in rng for b in rng for c in rng])
    far = [0, 1, 2, 3, 5, 60, 61, 62, 63, 64, 65, 100, 101, 102, 103, 1000, 1001, 2038, 2039, 2040]
    check([(a, b, c) for a in far for b in far for c in far])


def test_sycamore53_supremacy_sequence():
    """BASELINE config 5's network (tnco_amd/synthetic.py; the reference ships none -- tnco/utils/circuit.py would build it
    from cirq): 53 qubits, 86 couplers in four patterns with A, B the two staggered halves of ONE orientation and C, D
    those of the other (cirq's GRID_STAGGERED_PATTERN), ABCDCDAB -> 430 two-qubit gates at depth 20, the published count;
    the easier assignment of rounds 1-3 (the orientations alternate per cycle) is kept for the fixtures made on it."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.sycamore53_tn(20)
    gates = [t for t in ts if len(t) == 4]
    assert (len(ts), len(gates), dims, out) == (536, 430, 2, ()) and sum(len(t) == 1 for t in ts) == 106
    n_inds = 1 + max(i for t in ts for i in t)
    assert n_inds == 913 and all(sum(i in t for t in ts) == 2 for i in (0, 500, 912))
    # the patterns: rebuild them as the generator does and look at their orientation
    names = syn._SYCAMORE_LAYOUTS["supremacy"]
    assert {names[(1, -1)], names[(0, -1)]} == {"A", "B"} and {names[(0, +1)], names[(1, +1)]} == {"C", "D"}
    alt = syn._SYCAMORE_LAYOUTS["alternating"]
    assert {alt[(0, +1)], alt[(0, -1)]} == {"A", "B"}  # (A, B of the old assignment: BOTH orientations from even rows)
    per_cycle = [len([t for t in syn.sycamore53_tn(d)[0] if len(t) == 4]) for d in range(1, 9)]
    steps = [b - a for a, b in zip([0] + per_cycle, per_cycle)]  # gates of cycles A B C D C D A B
    assert steps[0] == steps[6] and steps[1] == steps[7] and steps[2] == steps[4] and steps[3] == steps[5]
    assert steps[0] + steps[1] + steps[2] + steps[3] == 86
    # four cycles connect the lattice (two alone, one orientation, do not)
    from tnco_amd.app.tn import get_connected_components
    assert len(get_connected_components(syn.sycamore53_tn(4)[0])) == 1
    assert len(get_connected_components(syn.sycamore53_tn(2)[0])) > 1
    old = syn.sycamore53_tn(20, "alternating")[0]
    assert (len(old), sum(len(t) == 4 for t in old)) == (541, 435)
    # the CZ variant: diagonal gates decomposed into hyper-indices (wire segments on two or three tensors), fused or raw
    raw = syn.sycamore53_cz_tn(12, fuse=None)[0]
    holders = {}
    for t in raw:
        for i in t:
            holders[i] = holders.get(i, 0) + 1
    assert len(raw) == 1000 and set(holders.values()) == {2, 3}
    fused = syn.sycamore53_cz_tn(12, fuse=4, seed=0)[0]
    assert len(fused) < len(raw) / 3 and max(len(t) for t in fused) <= 4 and len(get_connected_components(fused)) == 1


def test_replica_seeds_through_numpy_are_random_choices():
    """_sa_driver.replica_seeds: the per-run seeds of sa.py:237 (`Random(seed).choices(range(2**32), k=n_runs)`)
    drawn through numpy's legacy MT19937 from the same state -- same seeds, same generator state afterwards."""
    import random

    from tnco_amd.app._sa_driver import replica_seeds
    for seed, k, skip in [(0, 65536, 0), (1, 4096, 5), (12345, 70001, 623), (2**40 + 7, 5000, 1300), (3, 100, 0)]:
        a, b = random.Random(seed), random.Random(seed)
        for _ in range(skip):
            a.random(), b.random()
        assert a.choices(range(2**32), k=k) == replica_seeds(b, k)
        assert a.getstate() == b.getstate() and a.random() == b.random()


def test_graph_form_of_the_greedy_equals_the_set_form():
    """The claim greedy_graph_kernel rests on (csrc/greedy_graph.h), on the CPU: without hyper-indices opt_einsum's
    greedy over index sets is a greedy over a multigraph -- ssa ids, leg counts, (neighbour, shared legs) lists,
    a union-find over dead ids.  tools/greedy_graph_model.py against ctree.ssa_greedy, path for path."""
    import pathlib
    import sys
    from random import Random

    import numpy as np

    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tools"))
    import greedy_graph_model as gm
    from tests.test_gpu_greedy import _hubs, _random_multigraph
    from tnco_amd import ctree as ct, synthetic as syn

    rng = np.random.default_rng(7)
    nets = [_random_multigraph(rng) for _ in range(40)]
    nets.append((*_hubs(30, double={(0, 3)}), []))
    reg = syn.regular_problem(64, 7)
    nets.append((reg.ts_inds, reg.n_inds, []))
    same = 0
    for ts, _n_inds, out in nets:
        for seed in (0, 5):
            order = list(range(len(ts)))
            Random(seed).shuffle(order)
            inputs = [frozenset(ts[t]) for t in order]
            ref = ct.ssa_greedy(inputs, frozenset(out))
            got = gm.graph_greedy(inputs, frozenset(out))
            if got is None:  # (a second component: the set form goes on with outer products)
                continue
            norm = lambda p: [(min(a, b), max(a, b)) for a, b in p]  # noqa: E731
            assert norm(got) == norm(ref)
            same += 1
    assert same > 60
