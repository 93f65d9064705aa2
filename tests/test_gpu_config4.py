"""BASELINE config 4 on what a one-GPU box can run: the 512-leaf network with 524 288 replicas -- the whole
8-GPU job of tnco/app/infinite_memory/sa.py:237-257 (n_runs seeds from ONE Random(seed).choices list, run by run
independent) -- on ONE device, its initial trees drawn on that device and never copied to the host.  What the
8-way launch adds on top (tnco/parallel.py:330-341 -> shard_bounds + one all-reduce(min)) is arithmetic on
the run list, covered on the CPU by tests/test_parallel_gloo.py::test_config4_shards_of_eight."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_c4_524288_replicas_on_one_device(oracle_lib):
    from tnco_amd import core, parallel
    prob = H.regular_problem(512, graph_seed=11)
    R, world = 524288, 8
    seeds = H.replica_seeds(R)
    sweeps = 50
    betas = H.linear_betas(0, 100, 400)[:sweeps]
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0, keep_on_device=True)
    assert isinstance(links, core.DeviceLinks) and links.shape == (R, 3, 1023)
    rng = np.random.RandomState(4)
    # 64 replicas spread over the batch: eight from every 65 536-run shard an 8-GPU launch would hand to a rank, incl.
    # the first and the last run of every shard
    sample = []
    for k in range(world):
        lo, hi = parallel.shard_bounds(R, world, k)
        assert hi - lo == 65536
        sample += [lo, hi - 1] + sorted(int(x) for x in rng.randint(lo + 1, hi - 1, 6))
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds) as gpu:
        core.greedy_release()
        start = gpu.trees(sample, which_min=False, contraction=False)[0]  # (the device-drawn starts of the sample)
        tot0, mn0 = gpu.costs()
        gpu.run(betas[:20])
        gpu.run(betas[20:])
        assert gpu.validate() == (0, -1)
        tot, mn = gpu.costs()
        assert np.all(np.isfinite(tot)) and np.all(mn <= tot) and np.all(mn <= mn0)
        assert int(gpu.moves_per_replica().min()) > 0
        for j, r in enumerate(sample):
            o = H.make_oracle(oracle_lib, prob, start[j], seeds[r])
            o.run(oracle_lib.PROB_MH, betas)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        # the head of the result list (k-select on the device) == the host's sort, ties to the lower run id
        c, ids = gpu.best(64)
        order = np.lexsort((np.arange(R), mn))[:64]
        assert np.array_equal(c, mn[order]) and np.array_equal(mn[ids], c)
        # what an 8-GPU launch reduces: the minimum over the shards' minima (one all-reduce(min)) is the batch minimum
        shard_min = [mn[slice(*parallel.shard_bounds(R, world, k))].min() for k in range(world)]
        assert min(shard_min) == c[0]
    # the starts are the reference's recipe whatever the batch they are drawn in: the first shard alone gives the same trees
    alone = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds[:65536], device=0)
    pick = [j for j, r in enumerate(sample) if r < 65536]
    assert np.array_equal(alone[[sample[j] for j in pick]], start[pick])
    core.greedy_release()
    core.release_cached()
