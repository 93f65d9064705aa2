"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, same
seeded inputs, bit-exact (integer tree structure, leg masks, PRNG state AND the
float64 costs: with dims = 2 every cost is an exact power of two and all other
operations are single IEEE add/sub/div; the only transcendental, pow() in the
Metropolis rule, can flip a decision with probability ~1e-16 per move)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def core():
    from tnco_amd import core as c
    return c


def _run_both(core, orc, prob, seeds, betas, prob_kind="mh", n_check=None, links=None, **kw):
    links = prob.links(seeds) if links is None else links
    okw = {k: v for k, v in kw.items() if k in ("cost_type", "disable_shared_inds", "n_projs")}
    gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=prob.dims,
                                output_mask=prob.output_mask, sparse_mask=prob.sparse_mask, **kw)
    gpu.run(betas, prob_kind)
    kind = {"base": 0, "greedy": 1, "mh": 2}[prob_kind]
    tot, mn = gpu.costs()
    moves = gpu.moves_per_replica()
    n_check = len(seeds) if n_check is None else n_check
    for r in range(n_check):
        o = H.make_oracle(orc, prob, links[r], seeds[r], **okw)
        o.run(kind, betas)
        H.assert_replica_equal(gpu, r, o)
        assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        assert int(moves[r]) == o.counters()["moves"]
    assert gpu.validate() == (0, -1)
    return gpu


def test_c2_64leaf_regular(core, oracle_lib):
    """BASELINE config 2: 64-leaf 3-regular TN, 4096 replicas, fixed seeds."""
    prob = H.regular_problem(64, graph_seed=7)
    seeds = H.replica_seeds(4096)
    betas = H.linear_betas(0, 100, 1000)
    gpu = _run_both(core, oracle_lib, prob, seeds, betas, n_check=256)
    c, ids = gpu.best(8)
    _, mn = gpu.costs()
    assert np.all(np.diff(c) >= 0) and c[0] == mn.min()


def test_c3_512leaf_regular(core, oracle_lib):
    """BASELINE config 3 topology (512 leaves, 768 indices, 12 words), fewer replicas."""
    prob = H.regular_problem(512, graph_seed=11)
    seeds = H.replica_seeds(256)
    betas = H.linear_betas(0, 100, 400)
    _run_both(core, oracle_lib, prob, seeds, betas, n_check=16)


@pytest.mark.parametrize("n,deg", [(4, 3), (8, 3), (16, 3), (40, 5), (130, 3), (200, 4)])
def test_sizes_and_lane_groups(core, oracle_lib, n, deg):
    """W = 1, 1, 1, 2, 4, 7 words -> groups of 1, 1, 1, 2, 4, 8 lanes."""
    prob = H.regular_problem(n, graph_seed=n, degree=deg)
    seeds = H.replica_seeds(70, S=n)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 50, 300))


@pytest.mark.parametrize("n,deg,R,sweeps", [(2, 1, 5, 20), (3, 2, 5, 30), (900, 3, 6, 60), (1300, 3, 5, 50),
                                            (1400, 3, 5, 50), (2600, 3, 4, 40)])
def test_edge_sizes_and_wide_masks(core, oracle_lib, n, deg, R, sweeps):
    """Smallest trees (no move / one move per sweep) and every wide lane layout: 22 words -> 8 lanes
    x 3, 31 -> 8 x 4, 33 -> 16 x 3, 61 -> 16 x 4."""
    prob = H.regular_problem(n, graph_seed=n % 97, degree=deg)
    seeds = H.replica_seeds(R, S=n)
    gpu = _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 80, sweeps))
    gpu.run([], "mh")  # zero sweeps: no-op
    assert gpu.validate() == (0, -1)


def test_too_many_indices_is_refused(core):
    prob = H.regular_problem(2800, graph_seed=3)  # 4200 indices > 4096
    seeds = H.replica_seeds(1)
    with pytest.raises(NotImplementedError):
        core.BatchedOptimizer(prob.leaf_masks, prob.links(seeds), seeds, n_inds=prob.n_inds)


@pytest.mark.parametrize("kind", ["base", "greedy", "mh"])
def test_prob_kinds(core, oracle_lib, kind):
    prob = H.regular_problem(48, graph_seed=3)
    seeds = H.replica_seeds(33, S=5)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0.5, 20, 200), prob_kind=kind)


def test_chunked_run_equals_single_run(core, oracle_lib):
    prob = H.regular_problem(64, graph_seed=7)
    seeds = H.replica_seeds(40, S=9)
    betas = H.linear_betas(0, 100, 300)
    links = prob.links(seeds)
    a = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
    b = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
    a.run(betas)
    for lo, hi in [(0, 1), (1, 17), (17, 18), (18, 300)]:
        b.run(betas[lo:hi])
    for r in range(len(seeds)):
        for x, y in zip(a.tree(r), b.tree(r)):
            assert np.array_equal(x, y)
        assert np.array_equal(a.prng_state(r), b.prng_state(r))
    assert np.array_equal(a.costs()[1], b.costs()[1])


def test_best_tree_journal_overflow(core, oracle_lib, monkeypatch):
    """min_ctree bookkeeping: more accepted rotations than the rotation log holds without an
    improvement (overflow -> full copy at the next improvement), then a long descent (log replay on
    read-back), across launches."""
    prob = H.regular_problem(96, graph_seed=21)
    seeds = H.replica_seeds(48, S=21)
    links = prob.links(seeds)
    monkeypatch.setenv("TNCO_HIP_JLOG_CAP", "256")
    gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
    monkeypatch.delenv("TNCO_HIP_JLOG_CAP")
    phases = [("base", np.zeros(60)), ("greedy", np.zeros(150)), ("base", np.zeros(40)),
              ("mh", H.linear_betas(1, 200, 300))]
    for kind, betas in phases:
        gpu.run(betas, kind)
    c = gpu.counters()
    assert c["full_copies"] > 0 and c["improved"] > c["full_copies"]
    for r in range(len(seeds)):
        o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
        for kind, betas in phases:
            o.run({"base": 0, "greedy": 1, "mh": 2}[kind], betas)
        H.assert_replica_equal(gpu, r, o)
    assert gpu.validate() == (0, -1)


def test_hyper_output_dims(core, oracle_lib):
    """Hyper-indices + output legs; uniform dims 3 (table path)."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(30, 70, k=4, n_output=6, seed=2)
    prob = H.Problem(ts, 3, out)
    seeds = H.replica_seeds(40, S=2)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 300))


def test_hyper_pow2_dims(core, oracle_lib):
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(40, 90, k=3, n_output=5, seed=4)
    prob = H.Problem(ts, 2, out)
    seeds = H.replica_seeds(40, S=3)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 300))


@pytest.mark.parametrize("n,n_inds,k", [(200, 470, 3), (512, 768, 3), (700, 1330, 4)])
def test_hyper_networks_with_blocks_longer_than_a_line(core, oracle_lib, n, n_inds, k):
    """Infinite memory + hyper-indices at 8 / 12 / 21 mask words (round 5: the hyper legs are not stored -- the sweep
    kernel derives hyper[p] = legs(p) & legs(c0) & legs(c1) from the own legs of B and A and the children's legs it
    carries -- and the caches the host reads back are derived the same way) against the oracle bit for bit: trees,
    legs, both caches (hyper legs included), generator -- with output legs, all three rules, validation."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(n, n_inds, k=k, n_output=7, seed=n)
    prob = H.Problem(ts, 2, out)
    seeds = H.replica_seeds(24, S=n)
    gpu = _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 40, 120), n_check=12)
    gpu.run(H.linear_betas(0, 5, 10), "greedy")
    gpu.run(np.zeros(5), "base")
    assert gpu.validate() == (0, -1)
    links = prob.links(seeds)
    for r in range(4):
        o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
        o.run(oracle_lib.PROB_MH, H.linear_betas(0, 40, 120))
        o.run(oracle_lib.PROB_GREEDY, H.linear_betas(0, 5, 10))
        o.run(oracle_lib.PROB_BASE, np.zeros(5))
        H.assert_replica_equal(gpu, r, o)


def test_vector_dims(core, oracle_lib):
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(24, 60, k=3, n_output=3, seed=5, dims_choices=(2, 3, 4, 7))
    prob = H.Problem(ts, np.array(dims, np.uint64), out)
    seeds = H.replica_seeds(20, S=4)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 200))


def test_vector_dims_one_odd_part(core, oracle_lib):
    """Per-index dims 2^a * m with ONE odd part m for all of them ({2, 3, 4, 6, 12}; also a large power
    of two times 3): the sequential product of simple.hpp:51-53 is a table of the number of odd factors
    times an exact power of two -- same bits as the oracle's loop, float64 and float32 cost, with
    sparse legs, and with the table switched off (the factor-by-factor chain over the odd parts)."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(26, 64, k=3, n_output=2, seed=9, dims_choices=(2, 3, 4, 6, 12, 3))
    dims = list(dims)
    dims[5] = 3 * 2**21
    prob = H.Problem(ts, np.array(dims, np.uint64), out)
    seeds = H.replica_seeds(20, S=6)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 200))
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 100), cost_type="float32")
    probs = H.Problem(ts, np.array(dims, np.uint64), out, sparse_inds=[1, 4, 11, 30, 31, 60])
    _run_both(core, oracle_lib, probs, seeds, H.linear_betas(0, 30, 100), n_projs=7)


def test_vector_dims_chain_without_the_table(core, oracle_lib, monkeypatch):
    monkeypatch.setenv("TNCO_HIP_NO_ODD_TABLE", "1")
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(26, 64, k=3, n_output=2, seed=9, dims_choices=(2, 3, 4, 6, 12, 3))
    prob = H.Problem(ts, np.array(dims, np.uint64), out)
    _run_both(core, oracle_lib, prob, H.replica_seeds(12, S=6), H.linear_betas(0, 30, 120))


def test_sparse_inds(core, oracle_lib):
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(24, 60, k=3, n_output=4, seed=6)
    prob = H.Problem(ts, 2, out, sparse_inds=[1, 5, 9, 20, 33, 47])
    seeds = H.replica_seeds(20, S=6)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 200), n_projs=5)


def test_float32_cost(core, oracle_lib):
    prob = H.regular_problem(32, graph_seed=8)
    seeds = H.replica_seeds(24, S=8)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 40, 300), cost_type="float32")


def test_disable_shared_inds(core, oracle_lib):
    prob = H.regular_problem(32, graph_seed=9)
    seeds = H.replica_seeds(24, S=10)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 40, 300), disable_shared_inds=True)


def test_prng_roundtrip_and_resume(core, oracle_lib):
    """prng_state export/import (optimize/optimizer.hpp:68-71,191-195): a replica
    restarted from its exported state continues identically."""
    prob = H.regular_problem(32, graph_seed=12)
    seeds = H.replica_seeds(8, S=12)
    links = prob.links(seeds)
    betas = H.linear_betas(0, 30, 120)
    a = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
    a.run(betas[:50])
    # rebuild b from a's trees + prng states, then run both to the end
    cur = np.stack([np.stack(a.tree(r)[:3]) for r in range(len(seeds))])
    b = core.BatchedOptimizer(prob.leaf_masks, cur, seeds, n_inds=prob.n_inds)
    for r in range(len(seeds)):
        b.set_prng_state(r, a.prng_state(r))
        assert np.array_equal(b.prng_state(r), a.prng_state(r))
    a.run(betas[50:])
    b.run(betas[50:])
    for r in range(len(seeds)):
        for x, y in zip(a.tree(r), b.tree(r)):
            assert np.array_equal(x, y)


def test_invalid_inputs(core):
    prob = H.regular_problem(16, graph_seed=1)
    seeds = H.replica_seeds(4)
    links = prob.links(seeds)
    bad = links.copy()
    bad[2, 2, 0] = 5  # wrong parent
    with pytest.raises(ValueError):
        core.BatchedOptimizer(prob.leaf_masks, bad, seeds, n_inds=prob.n_inds)
    with pytest.raises(NotImplementedError):
        core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, cost_type="float128")
    # a contraction of two tensors that share no index is rejected unless disable_shared_inds
    from tnco_amd import ctree as ct
    ts = [[0], [0, 1], [1, 2], [2]]
    lm = ct.pack_masks(ts, 3)
    l, r, p = ct.tree_from_contraction([(0, 3, 4), (1, 2, 5), (4, 5, 6)], 4)
    lk = np.stack([l, r, p])
    with pytest.raises(ValueError, match="Contraction is not valid"):
        core.BatchedOptimizer(lm, lk, [1], n_inds=3)
    core.BatchedOptimizer(lm, lk, [1], n_inds=3, disable_shared_inds=True).close()
    # per-index dims: a power-of-two part up to 2^32 is taken (the exponent classes live in LDS), a larger one refused
    dims = np.full(prob.n_inds, 2, np.uint64)
    dims[3] = 3 << 32
    g = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=dims)
    g.run(H.linear_betas(0, 10, 5))
    assert g.validate() == (0, -1)
    g.close()
    dims[3] = 1 << 40
    with pytest.raises(NotImplementedError, match="2\\^32"):
        core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=dims)


def test_few_small_trees_run_lds_resident(core, oracle_lib):
    """csrc/sa_small.h: a handle of small trees (<= 128 leaves, <= 2 mask words, the fast cost path; above 64 leaves: no
    more replicas than the CUs hold in two rounds) keeps every replica's tree in LDS during a launch -- launch_groups == 0
    says so.
    (a) Same bits as the oracle with launches of 1, 7 and 40 sweeps: 64 leaves (the 63-node instantiation), 84 leaves (the
    127-node one), 10 leaves, 2 leaves.  (b) The same replicas at the head of a batch too large to stay resident run through
    the HBM kernel (launch_groups >= 1) and end in the same state bit for bit: trees, best trees, costs, generator state."""
    for n, gs, R in ((64, 7, 96), (84, 11, 40), (10, 2, 70), (2, 1, 5)):
        prob = H.regular_problem(n, graph_seed=gs, degree=3 if n > 2 else 1)
        R_big = 20000 if n == 84 else 0
        seeds = H.replica_seeds(max(R, R_big), S=n)
        links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds) if R_big else prob.links(seeds)
        betas = H.linear_betas(0, 60, 48)
        gpu = core.BatchedOptimizer(prob.leaf_masks, links[:R], seeds[:R], n_inds=prob.n_inds)
        assert gpu.launch_groups == 0
        gpu.run(betas[:1]); gpu.run(betas[1:8]); gpu.run(betas[8:])
        tot, mn = gpu.costs()
        for r in range(0, R, 3):
            o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
            o.run(oracle_lib.PROB_MH, betas)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        assert gpu.validate() == (0, -1)
        if R_big:
            big = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
            assert big.launch_groups >= 1
            big.run(betas[:1]); big.run(betas[1:8]); big.run(betas[8:])
            tb, mb = big.costs()
            assert np.array_equal(tb[:R], tot) and np.array_equal(mb[:R], mn)
            assert np.array_equal(big.moves_per_replica()[:R], gpu.moves_per_replica())
            for r in range(R):
                for which in (False, True):
                    for x, y in zip(gpu.tree(r, which_min=which), big.tree(r, which_min=which)):
                        assert np.array_equal(x, y)
                assert np.array_equal(gpu.prng_state(r), big.prng_state(r))
            assert big.validate() == (0, -1)
            big.close()
        gpu.close()
    # not resident: a general cost model
    prob = H.regular_problem(64, graph_seed=7)
    seeds = H.replica_seeds(8)
    links = prob.links(seeds)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, cost_type="float32") as g:
        assert g.launch_groups == 1


@pytest.mark.parametrize("n,deg,R", [(130, 3, 12300), (200, 4, 12300), (512, 3, 12300), (900, 3, 8200), (1300, 3, 8200),
                                     (2048, 3, 4100), (512, 3, 3000), (512, 3, 6000), (900, 3, 3000), (2048, 3, 2500)])
def test_full_wavefronts_of_the_lane_layouts(core, oracle_lib, n, deg, R):
    """Small batches run the sweep kernel's SPREAD form or an LDS-resident kernel (round 5), so the parity tests above no
    longer reach the full wavefronts of `sa_run_kernel<L, K, ...>` that big batches run: here batches too large for either,
    4 x 1 / 4 x 2 / 4 x 3 / 8 x 3 / 8 x 4 / 16 x 3 lanes x words, the first, some middle and the last replicas against the
    oracle -- and the SPREAD form with four, eight, four and two replicas per wavefront (3 000 / 6 000 replicas of 512
    leaves, 3 000 of 900, 2 500 of 2 048)."""
    prob = H.regular_problem(n, graph_seed=n % 89, degree=deg)
    seeds = H.replica_seeds(R, S=n)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
    betas = H.linear_betas(0, 60, 12)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds) as gpu:
        assert gpu.launch_groups >= 1
        gpu.run(betas[:5]); gpu.run(betas[5:])
        tot, mn = gpu.costs()
        for r in [0, 1, 63, 64, R // 2, R - 65, R - 2, R - 1]:
            o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
            o.run(oracle_lib.PROB_MH, betas)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        assert gpu.validate() == (0, -1)


@pytest.mark.parametrize("kind", ["hyper", "dims 3", "hyper, dims 3", "float32"])
def test_full_wavefronts_of_the_other_instantiations(core, oracle_lib, kind):
    """... and the hyper-index / general-cost-model instantiations the same way: 12 300 replicas, eight of them against the oracle."""
    from tnco_amd import synthetic as syn
    if kind.startswith("hyper"):
        ts, _dims, out = syn.random_hyper_tn(120, 200, k=3, n_output=5, seed=12)
        prob = H.Problem(ts, 3 if "dims 3" in kind else 2, out)
    else:
        prob = H.regular_problem(128, graph_seed=5)
        if kind == "dims 3":
            prob = H.Problem(prob.ts_inds, 3, [])
    kw = dict(cost_type="float32") if kind == "float32" else {}
    R = 12300
    seeds = H.replica_seeds(R, S=77)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
    betas = H.linear_betas(0, 40, 12)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=prob.dims, output_mask=prob.output_mask,
                               **kw) as gpu:
        assert gpu.launch_groups >= 1
        gpu.run(betas[:5]); gpu.run(betas[5:])
        tot, mn = gpu.costs()
        for r in [0, 1, 63, 64, R // 2, R - 65, R - 2, R - 1]:
            o = H.make_oracle(oracle_lib, prob, links[r], seeds[r], **kw)
            o.run(oracle_lib.PROB_MH, betas)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        assert gpu.validate() == (0, -1)


@pytest.mark.parametrize("R", [1030, 2050, 4099, 16390])
def test_small_tree_batches_that_do_not_fill_their_last_wavefront(core, oracle_lib, R):
    """csrc/sa_small.h: a batch is spread over the chip's wavefront slots -- 1, 2, 4, 8 or 16 replicas per wavefront, the other
    lane groups shadowing them -- and the last wavefront may hold fewer: the first, some middle and the last replicas of
    1030 (two per wavefront), 2050 (four), 4099 (eight) and 16 390 (sixteen) against the oracle."""
    prob = H.regular_problem(64, graph_seed=7)
    seeds = H.replica_seeds(R, S=R)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
    betas = H.linear_betas(0, 60, 30)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds) as gpu:
        assert gpu.launch_groups == 0
        gpu.run(betas[:11]); gpu.run(betas[11:])
        tot, mn = gpu.costs()
        for r in [0, 1, 2, 3, 15, 16, 17, R // 2, R - 18, R - 17, R - 3, R - 2, R - 1]:
            o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
            o.run(oracle_lib.PROB_MH, betas)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        assert gpu.validate() == (0, -1)


@pytest.mark.parametrize("n,deg,R", [(100, 3, 48), (200, 4, 33), (256, 3, 40), (512, 3, 24), (680, 3, 9)])
def test_larger_trees_that_fit_the_lds_run_resident(core, oracle_lib, n, deg, R):
    """csrc/sa_small.h, sa_lds_kernel<K>: up to 16 mask words (4 lanes x K = 1 ... 4 words) a handle whose replicas all fit
    the CUs' LDS at once keeps them there during a launch (16-bit links, the leaf legs as index lists) -- the latency
    regime of the larger networks; a 512-leaf tree is 58 KiB.  Same bits as the oracle, launches of 1, 7 and 40 sweeps;
    and the same replicas in a batch too large for the LDS (the HBM kernel) end in the same state."""
    prob = H.regular_problem(n, graph_seed=n % 89, degree=deg)
    R_big = 3000 if n == 512 else 0
    seeds = H.replica_seeds(max(R, R_big), S=n)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds) if R_big else prob.links(seeds)
    betas = H.linear_betas(0, 60, 48)
    gpu = core.BatchedOptimizer(prob.leaf_masks, links[:R], seeds[:R], n_inds=prob.n_inds)
    assert gpu.launch_groups == 0
    gpu.run(betas[:1]); gpu.run(betas[1:8]); gpu.run(betas[8:])
    tot, mn = gpu.costs()
    for r in range(0, R, 4):
        o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
        o.run(oracle_lib.PROB_MH, betas)
        H.assert_replica_equal(gpu, r, o)
        assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
    assert gpu.validate() == (0, -1)
    if R_big:
        big = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
        assert big.launch_groups >= 1
        big.run(betas[:1]); big.run(betas[1:8]); big.run(betas[8:])
        tb, mb = big.costs()
        assert np.array_equal(tb[:R], tot) and np.array_equal(mb[:R], mn)
        for r in range(R):
            for which in (False, True):
                for x, y in zip(gpu.tree(r, which_min=which), big.tree(r, which_min=which)):
                    assert np.array_equal(x, y)
            assert np.array_equal(gpu.prng_state(r), big.prng_state(r))
        assert big.validate() == (0, -1)
        big.close()
    gpu.close()


def test_c3_full_size_properties(core, oracle_lib):
    """BASELINE config 3 at FULL size (512 leaves, 65 536 replicas): size-independent properties
    the reference's own tests assert (tests/test_utils.py:575-769) -- every replica is_valid() on the
    device (tree links, leg masks and both caches against a from-scratch rebuild), min <= cur, a
    greedy run never increases the cost, chunked launches == one launch -- plus bit-exact agreement
    of a sample of replicas spread over the whole batch with the oracle."""
    prob = H.regular_problem(512, graph_seed=11)
    R = 65536
    seeds = H.replica_seeds(R)
    links = core.random_trees(prob.ts_inds, prob.n_inds, seeds)
    betas = H.linear_betas(0, 100, 60)
    gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
    tot0, mn0 = gpu.costs()
    gpu.run(betas[:25])
    gpu.run(betas[25:])
    assert gpu.validate() == (0, -1)
    tot, mn = gpu.costs()
    assert np.all(mn <= tot) and np.all(mn <= mn0) and np.all(np.isfinite(tot))
    c, ids = gpu.best(16)
    assert np.all(np.diff(c) >= 0) and c[0] == mn.min() and mn[ids[0]] == c[0]
    for r in list(range(0, R, R // 24)) + [R - 1]:
        o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
        o.run(oracle_lib.PROB_MH, betas)
        H.assert_replica_equal(gpu, r, o)
        assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
    # one launch of the whole schedule on a fresh handle gives the same batch
    one = core.BatchedOptimizer(prob.leaf_masks, links[:4096], seeds[:4096], n_inds=prob.n_inds)
    one.run(betas)
    assert np.array_equal(one.costs()[0], tot[:4096]) and np.array_equal(one.costs()[1], mn[:4096])
    # greedy from here: the cost of no replica goes up
    gpu.run(np.zeros(10), "greedy")
    tot2, mn2 = gpu.costs()
    assert np.all(tot2 <= tot) and np.all(mn2 <= mn) and gpu.validate() == (0, -1)


def test_c3_full_size_from_the_reference_starts(core, oracle_lib):
    """BASELINE config 3 as bench.py runs it: 65 536 replicas from DEVICE-drawn greedy starts (the reference's
    recipe, tnco/utils/tn.py:189-230), stepped in chunks through the handle's two streams (the split of every
    step over two launches, tnco_hip_run): 64 replicas spread over both halves of the batch against the oracle,
    bit for bit, the best costs of 4 096 more, every replica valid."""
    prob = H.regular_problem(512, graph_seed=11)
    R = 65536
    seeds = H.replica_seeds(R)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0)
    betas = H.linear_betas(0, 100, 1200)[:150]
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds) as gpu:
        for lo in range(0, 150, 50):
            gpu.run(betas[lo:lo + 50])
        assert gpu.validate() == (0, -1)
        tot, mn = gpu.costs()
        rng = np.random.RandomState(3)
        for r in sorted(int(x) for x in rng.choice(R, 64, replace=False)) + [0, R // 2 - 1, R // 2, R - 1]:
            o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
            o.run(oracle_lib.PROB_MH, betas)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        _dt, _t, omn, _mv = oracle_lib.run_batch(links[30000:34096], prob.leaf_masks, seeds[30000:34096], betas,
                                                 n_inds=prob.n_inds, dims=2)
        assert np.array_equal(omn, mn[30000:34096])


def test_prng_state_string_is_libstdcxx_text(core):
    """`prng_state` (optimize/optimizer.hpp:191-195) is the text libstdc++ streams for the generator:
    compared with the strings the REAL std::mt19937 of the build image printed
    (tests/golden/stdlib_rng.json, oracle/stdlib_rng.cpp), after 0 draws, and restored from text."""
    import json
    import pathlib
    golden = json.loads((pathlib.Path(__file__).parent / "golden" / "stdlib_rng.json").read_text())
    cases = {c["seed"]: {s["draws"]: s["str"] for s in c["state"]} for c in golden["cases"] if c["state"]}
    assert set(cases) >= {0, 42}
    prob = H.regular_problem(16, graph_seed=1)
    seeds = [0, 42]
    links = prob.links(seeds)
    gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
    for r, s in enumerate(seeds):
        assert gpu.prng_state_string(r) == cases[s][0]
    # a state given as text (700 draws of seed 42) goes in and comes back unchanged
    gpu.set_prng_state(0, cases[42][700])
    assert gpu.prng_state_string(0) == cases[42][700]


def test_vector_dims_powers_of_two(core, oracle_lib):
    """Per-index dims that are all powers of two take the exponent-class path (one masked popcount
    per class instead of the reference's loop over the legs): same bits, f64 and f32, with and
    without sparse legs."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(28, 70, k=3, n_output=3, seed=8, dims_choices=(2, 4, 8, 2, 16))
    prob = H.Problem(ts, np.array(dims, np.uint64), out)
    seeds = H.replica_seeds(20, S=8)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 200))
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 100), cost_type="float32")
    probs = H.Problem(ts, np.array(dims, np.uint64), out, sparse_inds=[2, 3, 11, 30, 31, 60])
    _run_both(core, oracle_lib, probs, seeds, H.linear_betas(0, 30, 100), n_projs=6)


@pytest.mark.parametrize("left_deep", [True, False])
def test_caterpillar_trees_deeper_than_the_build_kernels_lds_stack(core, oracle_lib, left_deep):
    """create() from trees 239 levels deep (a chain network contracted end to end): build_kernel's traverse keeps
    160 stack entries per replica in LDS and spills the rest to the replica's scratch -- a left-deep spine needs
    ~480.  Caches, costs and a few sweeps bit for bit against the oracle; with finite width too (its own traverses)."""
    from tnco_amd.synthetic import Problem
    n = 240
    ts = [[i, i + 1] for i in range(n)]  # (indices 0 and n: the open ends of the chain)
    prob = Problem(ts, 2)
    N = 2 * n - 1
    left = np.full(N, -1, np.int32)
    right = np.full(N, -1, np.int32)
    parent = np.full(N, -1, np.int32)
    order = list(range(n)) if left_deep else list(range(n - 1, -1, -1))
    top = order[0]
    for k, leaf in enumerate(order[1:]):
        z = n + k
        # (left-deep: the spine is child 0 and the traverse's stack grows by two entries a level; else the spine is child 1)
        left[z], right[z] = (top, leaf) if left_deep else (leaf, top)
        parent[left[z]] = parent[right[z]] = z
        top = z
    tree = np.stack([left, right, parent])
    seeds = H.replica_seeds(6, S=11)
    links = np.repeat(tree[None], len(seeds), axis=0)
    _run_both(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 6), links=links)


def test_general_cost_model_on_22_mask_words(core, oracle_lib):
    """dims = 3 on a 900-leaf network: the cost-table kernels at 8 lanes x 3 words, full wavefronts."""
    prob = H.regular_problem(900, graph_seed=900 % 89, degree=3)
    prob.dims = 3
    R = 4200
    seeds = H.replica_seeds(R, S=900)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
    betas = H.linear_betas(0, 60, 10)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=3) as gpu:
        gpu.run(betas)
        tot, mn = gpu.costs()
        for r in [0, 63, R // 2, R - 1]:
            o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
            o.run(oracle_lib.PROB_MH, betas)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        assert gpu.validate() == (0, -1)
