"""Snapshot / restore of a whole batch across the C boundary: the constructor arguments that
Optimizer.__reduce__ round-trips in the reference (tnco/optimize/infinite_memory/optimizer.py:243-245,
finite_width/optimizer.py:343-346 over include/tnco/optimize/infinite_memory/optimizer.hpp:61-88,
finite_width/greedy/optimizer.hpp:72-115): current trees, min_ctree, prng_state [, slices, min_slices].

A batch is stopped mid-schedule, its handle destroyed, a new one created from the snapshot, and the
schedule finished: trees, best trees, costs, slices and PRNG states must equal (a) the uninterrupted
GPU run, (b) the uninterrupted oracle, (c) an oracle restored the same way.  Like the reference the
restore rebuilds the caches and takes min_total_cost = get_cost(min_ctree[, min_slices]); on these
networks (dims 2, costs far below 2^53) every sum is exact, so (a) holds bit for bit."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def core():
    from tnco_amd import core as c
    return c


def _state(opt):
    ids = np.arange(opt.n_replicas)
    tot, mn = opt.costs()
    st = dict(cur=opt.trees(ids, which_min=False, contraction=False)[0], best=opt.trees(ids, which_min=True, contraction=False)[0],
              tot=tot, mn=mn, prng=opt.prng_states())
    if opt.finite_width:
        st["slices"], st["min_slices"] = opt.slices_many(ids)
    return st


def _assert_same(a, b):
    for k in a:
        x, y = a[k], b[k]
        if x.dtype == np.float64:
            x, y = x.view(np.uint64), y.view(np.uint64)
        bad = np.nonzero(np.any((x != y).reshape(len(x), -1), axis=1))[0]
        assert len(bad) == 0, (k, bad[:8])


def test_infinite_memory_batch_survives_destroy_and_restore(core, oracle_lib):
    orc = oracle_lib
    prob = H.regular_problem(64, graph_seed=7)
    R = 4096
    seeds = H.replica_seeds(R, S=5)
    links = prob.links(seeds)
    betas = H.linear_betas(0, 100, 400)
    kw = dict(n_inds=prob.n_inds, dims=2)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, **kw) as full:
        full.run(betas)
        want = _state(full)
    a = core.BatchedOptimizer(prob.leaf_masks, links, seeds, **kw)
    a.run(betas[:170])
    snap = a.snapshot()
    mid = _state(a)
    a.close()  # the handle is gone: only the snapshot (host arrays) survives
    assert snap["steps_done"] == 170 and snap["prng_states"].shape == (R, 625)
    with core.BatchedOptimizer.restore(snap, prob.leaf_masks, **kw) as b:
        got = _state(b)
        _assert_same(mid, got)  # incl. min_total_cost = get_cost(min_ctree) == the best partial[root] seen
        assert b.validate() == (0, -1)
        b.run(betas[170:])
        _assert_same(want, _state(b))
        assert b.validate() == (0, -1)
        # the oracle, uninterrupted and restored the same way (16 replicas)
        for r in range(0, R, R // 16):
            o = H.make_oracle(orc, prob, links[r], seeds[r])
            o.run(orc.PROB_MH, betas[:170])
            l, rr, p, m = o.tree(False)
            o2 = orc.Oracle(l, rr, p, m, n_inds=prob.n_inds, dims=2, mt_state=o.prng_state(), min_tree=o.tree(True))
            assert o2.min_total_cost == mid["mn"][r] and o2.total_cost == mid["tot"][r]
            for oo in (o, o2):
                oo.run(orc.PROB_MH, betas[170:])
                H.assert_replica_equal(b, r, oo)
                assert oo.min_total_cost == want["mn"][r] and oo.total_cost == want["tot"][r]


def test_finite_width_batch_survives_destroy_and_restore(core, oracle_lib):
    """... with slices, min_slices and the sweep counter that decides the re-slicing sweeps
    (n % update_slices, tnco/app/finite_width/sa.py:228): stopped at a sweep that is not a multiple of it."""
    orc = oracle_lib
    prob = H.regular_problem(48, graph_seed=8)
    R = 4096
    seeds = H.replica_seeds(R, S=3)
    links = prob.links(seeds)
    betas = H.linear_betas(0, 60, 120)
    kw = dict(n_inds=prob.n_inds, dims=2, max_width=6)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, **kw) as full:
        full.run(betas, update_slices_every=10)
        want = _state(full)
        assert want["slices"].any() and (want["slices"] != want["min_slices"]).any()
    a = core.BatchedOptimizer(prob.leaf_masks, links, seeds, **kw)
    a.run(betas[:47], update_slices_every=10)
    snap = a.snapshot()
    mid = _state(a)
    a.close()
    with core.BatchedOptimizer.restore(snap, prob.leaf_masks, **kw) as b:
        _assert_same(mid, _state(b))
        assert b.validate() == (0, -1)
        b.run(betas[47:], update_slices_every=10)
        _assert_same(want, _state(b))
        assert b.validate() == (0, -1)
        for r in range(0, R, R // 16):
            o = H.make_oracle(orc, prob, links[r], seeds[r], max_width=6)
            o.run(orc.PROB_MH, betas[:47], update_slices_every=10)
            l, rr, p, m = o.tree(False)
            sl, msl = o.slices()
            o2 = orc.Oracle(l, rr, p, m, n_inds=prob.n_inds, dims=2, mt_state=o.prng_state(), max_width=6,
                            slices=sl, min_tree=o.tree(True), min_slices=msl)
            assert o2.min_total_cost == mid["mn"][r] and o2.total_cost == mid["tot"][r]
            for oo in (o, o2):
                # (Oracle.run counts its sweeps from 0 in every call: drive both sweep by sweep with the true n,
                #  as tnco/app/finite_width/sa.py:228 does)
                for n in range(47, len(betas)):
                    oo.update(orc.PROB_MH, betas[n], update_slices=(n % 10 == 0))
                H.assert_replica_equal(b, r, oo)
                assert oo.min_total_cost == want["mn"][r]
                assert all(np.array_equal(x, y) for x, y in zip(b.slices(r), oo.slices()))


def test_restore_arguments_are_checked(core):
    prob = H.regular_problem(16, graph_seed=1)
    seeds = H.replica_seeds(4)
    links = prob.links(seeds)
    kw = dict(n_inds=prob.n_inds, dims=2)
    bad = links.copy()
    bad[1, 2, 0] = 5  # wrong parent in a best tree
    with pytest.raises(ValueError):
        core.BatchedOptimizer(prob.leaf_masks, links, seeds, min_links=bad, **kw)
    st = np.zeros((4, 625), np.uint32)
    st[2, 624] = 700
    with pytest.raises(ValueError, match="prng position"):
        core.BatchedOptimizer(prob.leaf_masks, links, None, prng_states=st, **kw)
    # one best tree shared by all replicas; states instead of seeds
    st[2, 624] = 624
    with core.BatchedOptimizer(prob.leaf_masks, links, None, prng_states=st, min_links=links[0], **kw) as o:
        assert np.array_equal(o.prng_states(), st)
        for r in range(4):
            assert np.array_equal(np.stack(o.tree(r, which_min=True)[:3]), links[0])
        assert np.all(o.costs()[1] == o.costs()[1][0])
        o.run(H.linear_betas(0, 10, 20))
        assert o.validate() == (0, -1)
        # prng states of a subset, set and read back
        s2 = o.prng_states([3, 1])
        o.set_prng_states(s2[::-1].copy(), [3, 1])
        assert np.array_equal(o.prng_states([1, 3]), s2)


def test_device_blocks_of_a_destroyed_optimizer_serve_the_next_one():
    """csrc/dev_cache.h: destroy keeps the handle's blocks of >= 1 MB, the next create of the same shape takes
    them (no hipMalloc), the run on recycled -- dirty -- memory is bit-identical, release_cached() empties it."""
    import numpy as np

    from tnco_amd import core, synthetic as syn
    prob = syn.regular_problem(64, graph_seed=7)
    seeds = syn.replica_seeds(2048)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
    betas = syn.linear_betas(0, 60, 30)
    core.release_cached()

    def run(**kw):
        with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, **kw) as opt:
            opt.run(betas, **({"update_slices_every": 10} if kw else {}))
            tot, mn = opt.costs()
            return tot.copy(), mn.copy(), [x for r in (0, 7, 2047) for x in opt.tree(r, which_min=True, with_masks=False)]

    first = run()
    held = int(core._lib.load().tnco_hip_diag_cached_bytes())
    assert held > 2048 * 32 * 1024  # (the rotation logs alone are 128 KB per replica)
    other = run(max_width=14)        # another shape in between: its blocks join the cache, dirtying nothing it should not
    again = run()
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])
    assert all(np.array_equal(a, b) for a, b in zip(first[2], again[2]))
    assert np.array_equal(other[0], run(max_width=14)[0])
    assert core.release_cached() >= held and int(core._lib.load().tnco_hip_diag_cached_bytes()) == 0
