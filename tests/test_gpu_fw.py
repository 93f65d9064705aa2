"""GPU parity of the finite-width (memory-constrained) optimizer against the oracle's restatement of
finite_width::greedy::Optimizer (include/tnco/optimize/finite_width/greedy/optimizer.hpp:72-390):
initial greedy slicing (draws from the PRNG through std::shuffle), width-gated moves, re-slicing
every `update_slices` sweeps with full cost-cache rebuild, best tree + best slices."""
import numpy as np
import pytest

from tests import helpers as H
from tnco_amd import ctree as ct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def core():
    from tnco_amd import core as c
    return c


def _initial_max_width(prob, links_r):
    l, r, p = links_r
    inds = prob.node_masks(l, r)
    return max(len(ct.unpack_mask(m)) for m in inds)


def _check(core, orc, prob, seeds, betas, max_width, chunks, every=10, links=None, **kw):
    links = prob.links(seeds) if links is None else links
    gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=prob.dims,
                                output_mask=prob.output_mask, sparse_mask=prob.sparse_mask, max_width=max_width, **kw)
    lo = 0
    for c in chunks:
        gpu.run(betas[lo:lo + c], "mh", update_slices_every=every)
        lo += c
    assert lo == len(betas)
    tot, mn = gpu.costs()
    for r in range(len(seeds)):
        okw = {k: v for k, v in kw.items() if k in ("cost_type", "skip_slices", "slices", "width_type",
                                                     "max_number_new_slices", "n_projs", "disable_shared_inds")}
        o = H.make_oracle(orc, prob, links[r], seeds[r], max_width=max_width, **okw)
        o.run(orc.PROB_MH, betas, update_slices_every=every)
        assert o.is_valid() == 0
        H.assert_replica_equal(gpu, r, o)
        gs, gms = gpu.slices(r)
        os_, oms = o.slices()
        assert np.array_equal(gs, os_), f"replica {r} slices"
        assert np.array_equal(gms, oms), f"replica {r} min_slices"
        assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
    # is_valid of every replica on the device (finite_width/greedy/optimizer.hpp:392-444): the
    # infinite-memory checks with the sliced cost cache, widths after slicing, the width cache
    assert gpu.validate() == (0, -1)
    return gpu


@pytest.mark.parametrize("wave", [True, False])
@pytest.mark.parametrize("n,frac", [(24, 0.5), (48, 0.6), (64, 0.4)])
def test_fw_regular(core, oracle_lib, monkeypatch, n, frac, wave):
    """Both forms of the re-slice: one wavefront per replica with the cost cache re-priced from the old costs
    (fw_wave_kernel: uniform power-of-two dims, pinned here -- the library would leave the form while many replicas
    fall back) and the general one, walk + full rebuild from the legs (TNCO_HIP_FW_WAVE=0; what hyper-indices,
    per-index dims, sparse legs and float32 costs always get)."""
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1" if wave else "0")
    prob = H.regular_problem(n, graph_seed=n + 1)
    seeds = H.replica_seeds(24, S=n)
    w0 = _initial_max_width(prob, prob.tree(seeds[0]))
    max_width = max(2, int(w0 * frac))
    betas = H.linear_betas(0, 60, 120)
    _check(core, oracle_lib, prob, seeds, betas, max_width, chunks=[120])


def test_fw_chunked_launches_keep_the_reslice_phase(core, oracle_lib):
    prob = H.regular_problem(40, graph_seed=5)
    seeds = H.replica_seeds(12, S=5)
    betas = H.linear_betas(0, 60, 95)
    _check(core, oracle_lib, prob, seeds, betas, 6, chunks=[1, 9, 10, 33, 42], every=7)


def test_fw_hyper_output_and_skip(core, oracle_lib):
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(30, 64, k=3, n_output=4, seed=3)
    prob = H.Problem(ts, 2, out)
    seeds = H.replica_seeds(10, S=3)
    skip = ct.pack_masks([[0, 5, 9, 11, 30, 41]], prob.n_inds)[0]
    betas = H.linear_betas(0, 40, 80)
    _check(core, oracle_lib, prob, seeds, betas, 7, chunks=[80], skip_slices=skip)


def test_fw_no_slicing_needed_equals_loose_bound(core, oracle_lib):
    """A bound no tensor reaches: no slices, every move is evaluated."""
    prob = H.regular_problem(24, graph_seed=2)
    seeds = H.replica_seeds(6, S=2)
    gpu = _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 60), 60, chunks=[60])
    assert not gpu.slices(0)[0].any()


def test_fw_unsupported(core):
    prob = H.regular_problem(16, graph_seed=1)
    seeds = H.replica_seeds(2)
    links = prob.links(seeds)
    with pytest.raises(RuntimeError):
        core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, max_width=-1)
    with pytest.raises(NotImplementedError):
        core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, max_width=4, width_type="float16")


def test_fw_float64_width(core, oracle_lib):
    prob = H.regular_problem(40, graph_seed=8)
    seeds = H.replica_seeds(10, S=8)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 60, 90), 6, chunks=[40, 50], width_type="float64")


@pytest.mark.parametrize("width_type", ["float32", "float64"])
def test_fw_per_index_dims(core, oracle_lib, width_type):
    """finite_width/cost_model/simple.hpp:48-56: width = running sum (in width_type) of log2(dims[p]);
    the greedy slicer breaks ties on log2(dims) (greedy/utils.hpp:50-60)."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(28, 60, k=3, n_output=3, seed=7, dims_choices=(2, 3, 4, 6))
    prob = H.Problem(ts, np.array(dims, np.uint64), out)
    seeds = H.replica_seeds(10, S=7)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 40, 80), 9.5, chunks=[80], width_type=width_type)


def test_fw_sparse_inds(core, oracle_lib):
    """finite_width/cost_model/simple_sparse_inds.hpp: width(inds - S) + min(width(inds & S), log2(n_projs))."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(28, 64, k=3, n_output=4, seed=9)
    prob = H.Problem(ts, 2, out, sparse_inds=[1, 5, 9, 20, 33, 47, 50, 51, 52])
    seeds = H.replica_seeds(10, S=9)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 40, 80), 6, chunks=[30, 50], n_projs=5)


def test_fw_sparse_inds_per_index_dims(core, oracle_lib):
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(24, 56, k=3, n_output=2, seed=10, dims_choices=(2, 3, 5))
    prob = H.Problem(ts, np.array(dims, np.uint64), out, sparse_inds=[0, 2, 7, 11, 19, 23, 40])
    seeds = H.replica_seeds(8, S=10)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 40, 70), 8.0, chunks=[70], n_projs=7)


@pytest.mark.parametrize("m", [1, 3])
def test_fw_max_number_new_slices(core, oracle_lib, m):
    """greedy/optimizer.hpp:226-321: a move that does not fit may slice up to m random further legs
    and is then tried against a full rebuild of the cost cache."""
    prob = H.regular_problem(36, graph_seed=12)
    seeds = H.replica_seeds(10, S=12)
    gpu = _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 50, 70), 5, chunks=[20, 50],
                 max_number_new_slices=m)
    assert gpu.counters()["accepted"] > 0


def test_sycamore53_depth20_finite_width(core, oracle_lib):
    """BASELINE config 5 topology (Sycamore-53 supremacy circuit, depth 20, pattern ABCDCDAB: 430 two-qubit
    gates, 536 tensors, 913 indices, 15 mask words -> 4 lanes x 4 words), memory-constrained, against the oracle."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.sycamore53_tn(20)
    prob = H.Problem(ts, 2, out)
    assert prob.n == 536 and prob.n_inds == 913 and sum(len(t) == 4 for t in ts) == 430
    seeds = H.replica_seeds(6, S=53)
    betas = H.linear_betas(0, 100, 60)
    gpu = _check(core, oracle_lib, prob, seeds, betas, 40, chunks=[25, 35], every=10)
    s, ms = gpu.slices(0)
    assert ms.any()


def test_sycamore53_infinite_memory(core, oracle_lib):
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.sycamore53_tn(20)
    prob = H.Problem(ts, 2, out)
    seeds = H.replica_seeds(32, S=54)
    links = prob.links(seeds)
    betas = H.linear_betas(0, 100, 300)
    gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds)
    gpu.run(betas)
    for r in range(8):
        o = H.make_oracle(oracle_lib, prob, links[r], seeds[r])
        o.run(oracle_lib.PROB_MH, betas)
        H.assert_replica_equal(gpu, r, o)
    assert gpu.validate() == (0, -1)


def test_fw_tensor_with_many_candidate_legs(core, oracle_lib):
    """A star with a 600-leg centre and a tiny max_width: 600 candidate legs in one shuffle (beyond the
    LDS fast path of the re-slice: the global-scratch path), bit-exact with the oracle."""
    ts = [list(range(600))] + [[i] for i in range(600)]
    prob = H.Problem(ts, 2)
    seeds = H.replica_seeds(3, S=1)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 20, 12), 10, chunks=[12], every=5)


@pytest.mark.parametrize("stack", ["0", "3"])
def test_fw_traversal_fallbacks(core, oracle_lib, monkeypatch, stack):
    """The walk over the tree without its LDS stack (TNCO_HIP_FW_STACK=0: the links are walked, what
    trees of more than 8192 nodes get) and with a 3-entry one (everything deeper goes through the
    global spill, what trees deeper than the LDS stack get): same results, bit for bit."""
    monkeypatch.setenv("TNCO_HIP_FW_STACK", stack)
    prob = H.regular_problem(64, graph_seed=3)
    seeds = H.replica_seeds(6, S=11)
    w0 = _initial_max_width(prob, prob.tree(seeds[0]))
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 60, 60), max(2, int(0.4 * w0)), chunks=[25, 35], every=5)
    prob = H.regular_problem(36, graph_seed=12)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 50, 40), 5, chunks=[40], max_number_new_slices=2)


def _spine_with_cherries(n, left_deep):
    """Pairs of leaves ('cherries') hanging off one long spine: the spine is the RIGHT child everywhere
    (left_deep=False: the backward walker of fw_walk2_kernel has a left child waiting at every level)
    or the LEFT child everywhere (the forward walker's stack is as deep as the spine)."""
    m = n // 2
    con, nxt = [], n
    cher = []
    for k in range(m):
        con.append((2 * k, 2 * k + 1, nxt)); cher.append(nxt); nxt += 1
    top = cher[-1]
    for k in reversed(range(m - 1)):
        con.append((top, cher[k], nxt) if left_deep else (cher[k], top, nxt)); top = nxt; nxt += 1
    return np.stack(ct.tree_from_contraction(con, n))


@pytest.mark.parametrize("left_deep", [False, True])
def test_fw_walk_from_both_ends_on_deep_trees(core, oracle_lib, left_deep):
    """fw_walk2_kernel beyond its LDS stacks: 64 levels of spine with a subtree waiting at every level --
    the backward walker's 40 entries overflow (it stops, the forward walker lists the rest) or the
    forward walker's stack spills to global memory; re-slicing every sweep, so the first walks see
    these trees nearly unchanged.  Bit for bit against the oracle."""
    n = 128
    m = n // 2
    # a ladder (rungs inside the cherries, rails between neighbouring cherries) + long-range bonds
    ts = [[] for _ in range(n)]
    nxt = 0
    for k in range(m):
        ts[2 * k].append(nxt); ts[2 * k + 1].append(nxt); nxt += 1              # rung
        if k + 1 < m:
            for side in (0, 1):
                ts[2 * k + side].append(nxt); ts[2 * (k + 1) + side].append(nxt); nxt += 1   # rails
        far = (k + 7) % m
        ts[2 * k].append(nxt); ts[2 * far + 1].append(nxt); nxt += 1            # long-range bond
    from tnco_amd.synthetic import Problem
    prob = Problem(ts, 2)
    seeds = H.replica_seeds(5, S=3)
    tree = _spine_with_cherries(n, left_deep)
    links = np.repeat(tree[None], len(seeds), axis=0)
    w0 = _initial_max_width(prob, tree)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 40, 12), max(3, w0 // 2), chunks=[12], every=1, links=links)


def test_fw_repricing_equals_the_full_rebuild_at_scale(core, monkeypatch):
    """The two forms of the re-slice's rebuild on the config-5 topology, 32 768 replicas x 30 sweeps from the
    reference's greedy starts: totals, best totals, slices and best slices identical.  (What the small
    cases above cannot show: events of one replica in ten thousand -- a re-slice that proposes the
    slices a replica already has must still re-associate the partial sums and may commit, as the
    reference's CostCache rebuild does; more than 32 changed indices early in a schedule.)"""
    from tnco_amd import synthetic as syn
    R = 32768
    p = syn.sycamore_problem(20)
    seeds = np.asarray(syn.replica_seeds(R))
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, device=0)
    betas = H.linear_betas(0, 100, 1200)[:30]
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1")
    a = core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=40)
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "0")
    b = core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=40)
    for c in range(0, 30, 10):
        monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1")
        a.run(betas[c:c + 10], update_slices_every=10)
        monkeypatch.setenv("TNCO_HIP_FW_WAVE", "0")
        b.run(betas[c:c + 10], update_slices_every=10)
        (ta, ma), (tb, mb) = a.costs(), b.costs()
        assert np.array_equal(ta, tb) and np.array_equal(ma, mb), c
    ids = np.arange(R)
    sa, sb = a.slices_many(ids), b.slices_many(ids)
    assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])
    assert a.validate() == (0, -1)


def test_fw_form_of_the_reslice_chosen_per_call(core, monkeypatch):
    """tnco_hip_run_fw switches between the two forms of the re-slice by the fall-backs of the previous call
    (random starts: many changed indices early -> the single kernel, later probes of the re-pricing):
    whatever it picks, the results are those of either form pinned."""
    from tnco_amd import synthetic as syn
    R = 4096
    p = syn.sycamore_problem(20)
    seeds = np.asarray(syn.replica_seeds(R))
    links = core.random_trees(p.ts_inds, p.n_inds, seeds)
    betas = H.linear_betas(0, 100, 1200)[:240]
    out = []
    for pin in (None, "0", "1"):
        if pin is None:
            monkeypatch.delenv("TNCO_HIP_FW_WAVE", raising=False)
        else:
            monkeypatch.setenv("TNCO_HIP_FW_WAVE", pin)
        o = core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=40)
        for c in range(0, 240, 20):  # twelve calls: the adaptive run changes form in between
            o.run(betas[c:c + 20], update_slices_every=10)
        out.append((o.costs(), o.slices_many(np.arange(R))))
        assert o.validate() == (0, -1)
        o.close()
    for (ca, sa) in out[1:]:
        assert np.array_equal(out[0][0][0], ca[0]) and np.array_equal(out[0][0][1], ca[1])
        assert np.array_equal(out[0][1][0], sa[0]) and np.array_equal(out[0][1][1], sa[1])


def test_config5_from_the_reference_starts_at_scale_against_the_oracle(core, oracle_lib, monkeypatch):
    """What bench.py's finite-width leg runs -- the config-5 topology from DEVICE-drawn greedy starts, the
    re-slice of a replica in one wavefront (fw_wave_kernel) -- against the oracle: 32 768
    replicas x 40 sweeps (re-slices at sweeps 0, 10, 20, 30), then every replica whose re-slice took a rare
    path at one of them (rebuilt in full although it has slices; more than 32 changed indices: the second
    pass over the paths) and 64 others, compared in full: trees, best trees, caches, slices, PRNG."""
    from tnco_amd import synthetic as syn
    orc = oracle_lib
    monkeypatch.delenv("TNCO_HIP_FW_WAVE", raising=False)
    R, MW = 32768, 32  # (max_width as in bench.py's finite-width leg)
    p = syn.sycamore_problem(20)
    assert p.n == 536 and p.n_inds == 913  # the supremacy sequence
    seeds = np.asarray(syn.replica_seeds(R))
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, device=0)
    betas = H.linear_betas(0, 100, 1200)[:40]
    rare = set()
    with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=MW) as gpu:
        for c in (1, 10, 10, 10, 9):  # every chunk but the last ends right after a re-slicing sweep
            lo = gpu._steps_done
            gpu.run(betas[lo:lo + c], update_slices_every=10)
            if lo + c < 40:
                how, nch = gpu.reslice_info()
                has = gpu.slices_many(np.arange(R))[0].any(axis=1)
                rare |= set(np.nonzero((how == 0) & has)[0].tolist())
                rare |= set(np.nonzero(nch > 32)[0].tolist())
                assert (nch >= 0).any()  # (the change lists were recorded: the walk-free form ran)
        assert gpu.validate() == (0, -1)
        rng = np.random.RandomState(5)
        ids = sorted(rare)[:96] + [int(x) for x in rng.choice(R, 64, replace=False)]
        tot, mn = gpu.costs()
        for r in ids:
            o = H.make_oracle(orc, p, links[r], seeds[r], max_width=MW)
            o.run(orc.PROB_MH, betas, update_slices_every=10)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
            assert all(np.array_equal(a, b) for a, b in zip(gpu.slices(r), o.slices()))
    print(f"rare-path replicas compared: {len(rare)} found, {min(len(rare), 96)} compared")


def test_config5_many_changed_indices_from_random_starts(core, oracle_lib, monkeypatch):
    """Random (Kruskal) starts change many indices per re-slice early on: with the re-pricing pinned, replicas
    with 33..64 changed indices (two passes over the paths) and with more (the full rebuild inside
    fw_reslice_b_kernel, its own traverse) are all there; each kind against the oracle."""
    from tnco_amd import synthetic as syn
    orc = oracle_lib
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1")
    R = 2048
    p = syn.sycamore_problem(20)
    seeds = np.asarray(syn.replica_seeds(R, S=9))
    links = core.random_trees(p.ts_inds, p.n_inds, seeds)
    betas = H.linear_betas(0, 100, 1200)[:21]
    kinds = {"<=32": set(), "33..64": set(), "rebuilt": set()}
    with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=40) as gpu:
        for c in (1, 10, 10):
            lo = gpu._steps_done
            gpu.run(betas[lo:lo + c], update_slices_every=10)
            how, nch = gpu.reslice_info()
            kinds["<=32"] |= set(np.nonzero((how == 1) & (nch >= 0) & (nch <= 32))[0].tolist())
            kinds["33..64"] |= set(np.nonzero((how == 1) & (nch > 32))[0].tolist())
            kinds["rebuilt"] |= set(np.nonzero(how == 0)[0].tolist())
        assert gpu.validate() == (0, -1)
        assert all(kinds.values()), {k: len(v) for k, v in kinds.items()}
        tot, mn = gpu.costs()
        for r in [x for v in kinds.values() for x in sorted(v)[:12]]:
            o = H.make_oracle(orc, p, links[r], seeds[r], max_width=40)
            o.run(orc.PROB_MH, betas, update_slices_every=10)
            H.assert_replica_equal(gpu, r, o)
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
            assert all(np.array_equal(a, b) for a, b in zip(gpu.slices(r), o.slices()))


def test_get_slices_per_wavefront_equals_the_lockstep_kernel(core, monkeypatch):
    """The re-slice of a replica in one wavefront (fw_wave_kernel: order | get_slices | re-pricing) against the general
    form (TNCO_HIP_FW_WAVE=0: fw_walk2_kernel | fw_reslice_kernel -- get_slices of sixteen replicas per wavefront in
    lock step, the cost cache rebuilt from the legs) on the config-5 network, 4 096 replicas x 60 sweeps -- some 5 000
    outputs of every generator, so the 624-word generations end inside shuffles -- and with its knobs turned so that
    the rare paths are the common ones: NO tensor's legs kept in LDS, all of them read from memory at every visit
    with the next group's request in flight (TNCO_HIP_FWS_CAP=0), and tensors with more than 44 candidate
    legs left to fw_reslice_a_kernel (TNCO_HIP_FWS_MAXNP=44: the two kernels share a launch).  Totals, best totals,
    slices, best slices and generator states identical."""
    from tnco_amd import synthetic as syn
    R = 4096
    p = syn.sycamore_problem(20)
    seeds = np.asarray(syn.replica_seeds(R, S=21))
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, device=0)
    betas = H.linear_betas(0, 100, 1200)[:60]
    ids = np.arange(R)

    def run(env):
        for k in ("TNCO_HIP_FW_WAVE", "TNCO_HIP_FWS_CAP", "TNCO_HIP_FWS_MAXNP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=40) as g:
            for c in range(0, 60, 20):
                g.run(betas[c:c + 20], update_slices_every=10)
            assert g.validate() == (0, -1)
            st = g.fw_stats()
            return g.costs(), g.slices_many(ids), np.asarray(g.prng_states()), st

    ref = run({"TNCO_HIP_FW_WAVE": "0"})
    assert ref[3]["repriced"] == 0 and ref[3]["full_rebuild_form"] == 6 * R
    for env in ({}, {"TNCO_HIP_FWS_CAP": "0"}, {"TNCO_HIP_FWS_CAP": "8"}, {"TNCO_HIP_FWS_MAXNP": "44"}):
        got = run(dict(env, TNCO_HIP_FW_WAVE="1"))
        assert got[3]["repriced"] == 6 * R and got[3]["full_rebuild_form"] == 0, env
        if "TNCO_HIP_FWS_MAXNP" in env:
            assert got[3]["too_many_wide"] > 0  # (replicas left to fw_reslice_a_kernel: the knob bites)
        assert np.array_equal(got[0][0], ref[0][0]) and np.array_equal(got[0][1], ref[0][1]), env
        assert np.array_equal(got[1][0], ref[1][0]) and np.array_equal(got[1][1], ref[1][1]), env
        assert np.array_equal(got[2], ref[2]), env


def test_fw_more_than_64_candidate_legs_in_one_wavefront(core, oracle_lib):
    """A 100-leg centre tensor, its 100 neighbours in a ring, max_width 70: the tensors around the centre have
    71..100 candidate legs -- two per lane of fw_wave_kernel, up to 50 variates drawn at once -- against the
    oracle."""
    m = 100
    ts = [list(range(m))] + [[i, m + i, m + (i + 1) % m] for i in range(m)]
    prob = H.Problem(ts, 2)
    seeds = H.replica_seeds(16, S=4)
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 24), 70, chunks=[24], every=3)


@pytest.mark.parametrize("wave", [True, False])
def test_fw_skip_slices_and_initial_slices_on_the_one_wavefront_path(core, oracle_lib, monkeypatch, wave):
    """`skip_slices` (indices get_slices must not slice) and caller-given initial `slices` on a network the
    one-wavefront re-slice takes (uniform dims 2, 96 tensors): fw_wave_kernel, and the general form, against the
    oracle."""
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1" if wave else "0")
    prob = H.regular_problem(96, graph_seed=5)
    seeds = H.replica_seeds(12, S=5)
    w0 = _initial_max_width(prob, prob.tree(seeds[0]))
    skip = ct.pack_masks([list(range(0, prob.n_inds, 3))], prob.n_inds)[0]
    init = ct.pack_masks([[1, 2, 4, 5, 7, 8, 10, 11]], prob.n_inds)[0]
    _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 50, 60), max(3, w0 // 2), chunks=[25, 35], every=5,
           skip_slices=skip, slices=init)


@pytest.mark.parametrize("n,degree,words", [(720, 3, 17), (900, 5, 36)])
def test_fw_one_wavefront_reslice_on_wide_networks(core, oracle_lib, monkeypatch, n, degree, words):
    """Networks of more than 16 mask words on the one-wavefront re-slice: a mask in 32 lanes (1 080 indices, two
    tensors per load) and in 64 lanes (2 250 indices, one) -- from the reference's greedy starts (random trees of
    the second network are wider than a double's exponent allows), against the oracle, and against the general
    form (TNCO_HIP_FW_WAVE=0) on more replicas."""
    prob = H.regular_problem(n, graph_seed=n + degree, degree=degree)
    assert (prob.n_inds + 63) // 64 == words
    seeds = np.asarray(H.replica_seeds(512, S=n))
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0)
    w0 = _initial_max_width(prob, links[0])
    mw = max(3, int(w0 * 0.6))
    betas = H.linear_betas(0, 40, 24)
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1")
    _check(core, oracle_lib, prob, seeds[:4], betas, mw, chunks=[24], every=4, links=links[:4])
    out = []
    for off in (False, True):
        monkeypatch.setenv("TNCO_HIP_FW_WAVE", "0" if off else "1")
        with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, max_width=mw) as g:
            g.run(betas, update_slices_every=4)
            assert g.validate() == (0, -1)
            out.append((g.costs(), g.slices_many(np.arange(len(seeds))), np.asarray(g.prng_states())))
    a, b = out
    assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1])
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("big", ["0", "1", None])
def test_fw_more_too_wide_tensors_than_the_one_wavefront_reslice_lists(core, oracle_lib, monkeypatch, big):
    """400 tensors under a bound nearly every contraction exceeds: some 390 too-wide tensors per replica, more than
    the 255 the lean configuration of fw_wave_kernel lists -- with the form pinned (TNCO_HIP_FW_WAVE=1: the library
    would leave it).  TNCO_HIP_FW_BIG=0: every replica leaves that kernel for the traverse of fw_reslice_a_kernel and
    the full rebuild of fw_reslice_b_kernel; =1: the roomier configuration (1 023 tensors, ten count planes) takes
    them; unset: the library switches to it after the first call.  Against the oracle each time."""
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1")
    if big is None:
        monkeypatch.delenv("TNCO_HIP_FW_BIG", raising=False)
    else:
        monkeypatch.setenv("TNCO_HIP_FW_BIG", big)
    prob = H.regular_problem(400, graph_seed=9)
    seeds = H.replica_seeds(6, S=9)
    gpu = _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 12), 4, chunks=[5, 7], every=3)
    how, nch = gpu.reslice_info()
    st = gpu.fw_stats()
    if big == "0":
        assert (how == 0).all() and st["too_many_wide"] == st["repriced"]  # (rebuilt in full: none was re-priced)
    elif big == "1":  # (the few left: a too-wide tensor more than 64 levels below the root -- the ordering's keys are 64 bits)
        assert st["too_many_wide"] < st["repriced"] and (how == 1).any()
    else:
        assert 0 < st["too_many_wide"] < st["repriced"] and (how == 1).any()  # (the first call's re-slices only)


def test_fw_more_than_128_candidate_legs_on_the_one_wavefront_path(core, oracle_lib, monkeypatch):
    """A 300-spoke hub made of three 100-spoke tensors (no LEAF is too wide) with the 300 rim tensors in a ring,
    max_width 150: the contractions around the hub have 150..300 candidate legs -- beyond the 128 the parallel shuffle
    of fw_wave_kernel draws at once; its roomier configuration lists up to 512 and runs the sequential shuffle for
    those.  Against the oracle."""
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1")
    monkeypatch.setenv("TNCO_HIP_FW_BIG", "1")
    m = 300
    hub = [list(range(100 * j, 100 * j + 100)) for j in range(3)]
    hub[0].append(2 * m); hub[1] += [2 * m, 2 * m + 1]; hub[2].append(2 * m + 1)
    ts = hub + [[i, m + i, m + (i + 1) % m] for i in range(m)]
    prob = H.Problem(ts, 2)
    seeds = H.replica_seeds(8, S=4)
    gpu = _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 30, 12), 150, chunks=[12], every=3)
    st = gpu.fw_stats()
    assert st["repriced"] > 0 and st["fell_back"] < st["repriced"]


@pytest.mark.parametrize("wave", [True, False])
@pytest.mark.parametrize("net", ["random_k4", "cz_fused", "cz_raw"])
def test_fw_hyper_index_networks_both_forms(core, oracle_lib, monkeypatch, net, wave):
    """Networks with hyper-indices -- what the reference's loader makes of diagonal gates (tnco/utils/tn.py:827, on by
    default) -- through both forms of the re-slice against the oracle: a random hypergraph with indices on up to four
    tensors and output indices, and the Sycamore lattice with CZ gates (wire segments held by two or three tensors),
    pre-fused as the loader does (fuse = 4) and raw.  In the one-wavefront form an index is among a node's children's
    legs while some, but not all, of its holders are below the node: marks up from every holder, then down from the
    root to where the paths meet."""
    from tnco_amd import synthetic as syn
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1" if wave else "0")
    if net == "random_k4":
        ts, dims, out = syn.random_hyper_tn(60, 130, k=4, n_output=3, seed=12)
    elif net == "cz_fused":
        ts, dims, out = syn.sycamore53_cz_tn(6, fuse=4, seed=3)
    else:
        ts, dims, out = syn.sycamore53_cz_tn(4, fuse=None)  # (four cycles: both coupler orientations, a connected network)
    prob = H.Problem(ts, 2, out)
    assert any(len(h) > 2 for h in prob.holders)
    seeds = H.replica_seeds(12, S=31)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, np.asarray(seeds), output_mask=prob.output_mask)
    w0 = _initial_max_width(prob, links[0])
    gpu = _check(core, oracle_lib, prob, seeds, H.linear_betas(0, 50, 60), max(3, int(0.7 * w0)), chunks=[25, 35], every=5, links=links)
    st = gpu.fw_stats()
    assert (st["repriced"] > 0) == wave and (st["full_rebuild_form"] > 0) == (not wave)
    if wave:
        assert st["fell_back"] < st["repriced"]  # (the re-pricing did take most of them)


def test_fw_hyper_index_network_forms_agree_at_scale(core, monkeypatch):
    """The CZ circuit network (depth 12, fused: ~220 tensors, 40 % of the indices on three tensors), 8 192 replicas x 60
    sweeps from greedy starts: the one-wavefront form and the general form end at identical totals, best totals,
    slices and generator states."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.sycamore53_cz_tn(12, fuse=4, seed=0)
    p = syn.Problem(ts, 2, out)
    R = 8192
    seeds = np.asarray(syn.replica_seeds(R, S=77))
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, device=0)
    w0 = _initial_max_width(p, links[0])
    betas = H.linear_betas(0, 100, 600)[:60]
    res = []
    for pin in ("1", "0"):
        monkeypatch.setenv("TNCO_HIP_FW_WAVE", pin)
        with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=int(0.7 * w0)) as g:
            for c in range(0, 60, 20):
                g.run(betas[c:c + 20], update_slices_every=10)
            assert g.validate() == (0, -1)
            res.append((g.costs(), g.slices_many(np.arange(R)), np.asarray(g.prng_states()), g.fw_stats()))
    a, b = res
    assert a[3]["repriced"] == 6 * R and b[3]["full_rebuild_form"] == 6 * R
    assert a[3]["fell_back"] < 0.05 * a[3]["repriced"], a[3]
    assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1])
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[2], b[2])


def test_fw_more_than_1024_tensors_on_the_one_wavefront_path(core, oracle_lib, monkeypatch):
    """The raw CZ circuit at depth 14: 1 149 tensors (24 or 32 nodes per lane of fw_wave_kernel, 19 KB of node table),
    three quarters of its indices on three tensors -- against the oracle, and the two forms of the re-slice against
    each other on 1 024 replicas."""
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.sycamore53_cz_tn(14, fuse=None)
    p = H.Problem(ts, 2, out)
    assert 1024 < p.n <= 2048
    seeds = np.asarray(H.replica_seeds(1024, S=14))
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, device=0)
    w0 = _initial_max_width(p, links[0])
    mw = int(0.7 * w0)
    monkeypatch.setenv("TNCO_HIP_FW_WAVE", "1")
    gpu = _check(core, oracle_lib, p, seeds[:4], H.linear_betas(0, 40, 30), mw, chunks=[30], every=5, links=links[:4])
    assert gpu.fw_stats()["repriced"] > 0
    betas = H.linear_betas(0, 60, 40)
    res = []
    for pin in ("1", "0"):
        monkeypatch.setenv("TNCO_HIP_FW_WAVE", pin)
        with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, max_width=mw) as g:
            g.run(betas, update_slices_every=10)
            assert g.validate() == (0, -1)
            res.append((g.costs(), g.slices_many(np.arange(len(seeds))), np.asarray(g.prng_states())))
    a, b = res
    assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1])
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("R", [2100, 5000, 8400])
@pytest.mark.parametrize("n,deg,frac", [(48, 3, 0.6), (200, 4, 0.7), (520, 3, 0.8), (900, 3, 0.85)])
def test_fw_full_wavefronts_of_the_staged_moves(core, oracle_lib, n, deg, frac, R):
    """Small finite-width batches run the staged moves with few replicas per wavefront (sa_sweep.h, SPREAD; round 5), so the
    tests above no longer reach their full wavefronts for layouts other than config 5's: 8 400 replicas (too many to spread:
    two halves on two streams), 4 x 1 / 4 x 2 / 4 x 4 / 8 x 3 lanes x words, the first, some middle and the last replicas
    against the oracle incl. their slices -- and 2 100 / 5 000 replicas: four / eight replicas per wavefront."""
    prob = H.regular_problem(n, graph_seed=n % 97, degree=deg)
    seeds = H.replica_seeds(R, S=n)
    links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
    max_width = max(2, int(_initial_max_width(prob, links[0]) * frac))
    betas = H.linear_betas(0, 60, 25)
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, max_width=max_width) as gpu:
        assert gpu.launch_groups == 2 or R < 8400  # (8 400: two halves in full wavefronts; smaller: spread, or halves at 8 x 3)
        gpu.run(betas, "mh", update_slices_every=10)
        tot, mn = gpu.costs()
        for r in [0, 1, 63, 64, R // 2 - 1, R // 2, R - 2, R - 1]:
            o = H.make_oracle(oracle_lib, prob, links[r], seeds[r], max_width=max_width)
            o.run(oracle_lib.PROB_MH, betas, update_slices_every=10)
            H.assert_replica_equal(gpu, r, o)
            assert all(np.array_equal(a, b) for a, b in zip(gpu.slices(r), o.slices()))
            assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        assert gpu.validate() == (0, -1)
