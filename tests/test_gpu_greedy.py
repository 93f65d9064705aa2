"""The device-side generator of the reference's initial trees (csrc/greedy_device.hip:
Random(seed).shuffle + opt_einsum's greedy, tnco/utils/tn.py:189-230) against the host generator
(csrc/host_greedy.cpp, itself tested equal to the Python spec in tests/test_host.py): the same trees,
tree for tree, whatever the network -- plain graphs, hyper-indices, output legs, equal index sets,
several components sharing one generator, networks outside the kernel's limits."""
import numpy as np
import pytest

from tnco_amd import core, ctree as ct, synthetic as syn

pytestmark = pytest.mark.gpu


def _both(ts, n_inds, seeds, out_keep=(), draws=None, on_device=True):
    """host == device; `on_device`: the kernel did every tree itself (none handed back to the host)."""
    from tnco_amd import _lib
    om = ct.pack_masks([list(out_keep)], n_inds)[0]
    d_host = None if draws is None else draws.copy()
    d_dev = None if draws is None else draws.copy()
    host = core.greedy_trees(ts, n_inds, seeds, output_mask=om, draws=d_host)
    dev = core.greedy_trees(ts, n_inds, seeds, output_mask=om, draws=d_dev, device=0)
    redone = _lib.load().tnco_hip_diag_greedy_device_redone()
    assert (redone == 0) if on_device else (redone != 0), redone
    bad = [k for k in range(len(seeds)) if not np.array_equal(host[k], dev[k])]
    assert not bad, (len(ts), bad[:8])
    if draws is not None:
        assert np.array_equal(d_host, d_dev)
        draws[:] = d_host
    return host


def _supported(ts, n_inds):
    from tnco_amd import _lib
    off, _hold = core.holders_csr(ts, n_inds)
    return bool(_lib.load().tnco_hip_diag_greedy_device_supported(len(ts), n_inds, off.ctypes.data))


@pytest.fixture(params=["graph", "set", "set-lds-queue"])
def form(request, monkeypatch):
    """The device forms of the greedy (the multigraph in LDS where the network has no hyper-index -- the
    default -- and the index sets in memory, with the candidate queue in registers or in LDS) with both
    shuffle kernels (state in LDS / in memory)."""
    if request.param != "graph":
        monkeypatch.setenv("TNCO_HIP_GREEDY_GRAPH", "0")
        monkeypatch.setenv("TNCO_HIP_SHUFFLE_LDS", "0")
    if request.param == "set-lds-queue":
        monkeypatch.setenv("TNCO_HIP_GREEDY_LDS_QUEUE", "1")
    return request.param


@pytest.mark.parametrize("n,gs", [(4, 1), (8, 1), (64, 7), (200, 3), (512, 11)])
def test_regular_graphs(n, gs, form):
    prob = syn.regular_problem(n, graph_seed=gs, degree=3 if n > 4 else 2)
    assert _supported(prob.ts_inds, prob.n_inds)
    seeds = np.concatenate([[0, 1, 42, 2**32 - 1, 123456789, 77], syn.replica_seeds(250)])
    links = _both(prob.ts_inds, prob.n_inds, seeds)
    for k in (0, 5, 100):  # every contraction shares an index (check_shared_inds, sa.py:186-190)
        ct.derive_inds(links[k, 0], links[k, 1], prob.leaf_masks, prob.output_mask, check_shared_inds=True)


@pytest.mark.parametrize("lds_queue", ["0", "1"])
def test_hyper_indices_outputs_equal_sets(lds_queue, monkeypatch):
    monkeypatch.setenv("TNCO_HIP_GREEDY_LDS_QUEUE", lds_queue)
    seeds = syn.replica_seeds(40)
    for seed in range(8):
        ts, _d, out = syn.random_hyper_tn(20, 37, k=3, n_output=4, seed=seed)
        n_inds = 1 + max(i for xs in ts for i in xs)
        cnt = [sum(i in xs for xs in ts) for i in range(n_inds)]
        assert _supported(ts, n_inds)
        _both(ts, n_inds, seeds, out_keep=[i for i in out if cnt[i] <= 1])
    _both([[0, 1], [0, 1], [1, 2], [2, 3], [1, 2], [3, 0]], 4, seeds)          # equal index sets
    _both([[0, 1, 9], [1, 2, 9], [2, 3, 9], [3, 4, 9], [4, 0, 9]], 10, seeds)  # an index common to all
    _both([[0, 1], [1, 2], [2, 3]], 4, seeds, out_keep=[0, 3])                  # the reference's docstring example


def test_fuzz_on_small_hyper_networks_against_the_python_spec():
    """The networks of tests/test_host.py's fuzz (equal index sets reappearing DURING the contraction:
    the case the round-2 advisor found wrong in both native generators), device == host == Python spec.
    Trees that end in outer products go to the host version inside the call: not asserted either way."""
    from tests.test_host import _small_hyper_networks
    seeds = np.array([0, 1, 2, 3, 5, 8, 13, 21])
    for ts, n_inds, out_keep in _small_hyper_networks(300):
        if not _supported(ts, n_inds):
            continue
        om = ct.pack_masks([list(out_keep)], n_inds)[0]
        dev = core.greedy_trees(ts, n_inds, seeds, output_mask=om, device=0)
        for k, sd in enumerate(seeds):
            con = ct.greedy_contraction(ts, out_keep, int(sd))
            assert np.array_equal(dev[k], np.stack(ct.tree_from_contraction(con, len(ts)))), (len(ts), int(sd))


def _hubs(spokes, ring=True, double=()):
    """Two hub tensors joined to each other and to every one of `spokes` small tensors (which form a ring):
    neighbour lists far longer than a wavefront, every neighbour of one hub a neighbour of the other."""
    ts = [[0], [0]] + [[] for _ in range(spokes)]
    nxt = 1
    for k in range(spokes):
        for hub in (0, 1):
            for _ in range(2 if (hub, k) in double else 1):
                ts[hub].append(nxt)
                ts[2 + k].append(nxt)
                nxt += 1
        if ring:
            ts[2 + k].append(nxt)
            ts[2 + (k + 1) % spokes].append(nxt)
            nxt += 1
    return ts, nxt


def test_graph_form_long_lists_output_and_dangling_legs():
    """What the regular graphs do not reach in the multigraph form: lists of several 64-entry pieces whose
    entries meet again across the pieces (two hubs with the same 100 neighbours), pairs of tensors sharing
    two legs, output legs, legs with a single holder; and its hand-backs: lists beyond 255 entries."""
    seeds = syn.replica_seeds(48)
    ts, n_inds = _hubs(100, double={(0, 3), (1, 3), (0, 50)})
    assert _supported(ts, n_inds)
    _both(ts, n_inds, seeds)
    # output legs on some tensors, dangling (single-holder, not output) legs on others
    ts2 = [list(x) for x in ts]
    out = []
    for k in range(0, 100, 7):
        ts2[2 + k].append(n_inds + len(out))
        out.append(n_inds + len(out))
    n2 = n_inds + len(out)
    for k in range(3, 100, 11):
        ts2[2 + k].append(n2)
        n2 += 1
    ts2[0].append(n2)
    n2 += 1
    _both(ts2, n2, seeds, out_keep=out)
    # a hub with 255 neighbours: its list and any other pass 255 entries -- every tree goes to the host inside the call
    from tnco_amd import _lib
    ts3 = [list(range(255))] + [[k, 255 + k, 255 + (k + 1) % 255] for k in range(255)]
    host = core.greedy_trees(ts3, 510, seeds[:8])
    dev = core.greedy_trees(ts3, 510, seeds[:8], device=0)
    assert np.array_equal(host, dev) and _lib.load().tnco_hip_diag_greedy_device_redone() == 8


def _random_multigraph(rng):
    """A connected network without hyper-indices: a random spanning tree + extra edges (some doubled or
    tripled), output legs and dangling legs on a few tensors; now and then a second component (its trees end
    in outer products: the host's, inside the call)."""
    n = int(rng.integers(3, 70))
    ts = [[] for _ in range(n)]
    nxt = 0

    def edge(a, b, m=1):
        nonlocal nxt
        for _ in range(m):
            ts[a].append(nxt)
            ts[b].append(nxt)
            nxt += 1
    split = int(rng.integers(2, n)) if (n > 6 and rng.random() < 0.1) else n
    for t in range(1, n):
        if t == split:
            continue  # (t starts a second component)
        lo = 0 if t < split else split
        edge(int(rng.integers(lo, t)), t, int(rng.choice([1, 1, 1, 2, 3])))
    for _ in range(int(rng.integers(0, 2 * n))):
        a, b = (int(x) for x in rng.integers(0, n, 2))
        if a != b and (a < split) == (b < split):
            edge(a, b, int(rng.choice([1, 1, 2])))
    out = []
    for t in range(n):
        if rng.random() < 0.15:
            ts[t].append(nxt)
            out.append(nxt)
            nxt += 1
        if rng.random() < 0.1:
            ts[t].append(nxt)  # a leg nobody else holds and that is no output leg
            nxt += 1
    return ts, nxt, out


def test_fuzz_graph_form_on_random_multigraphs():
    rng = np.random.default_rng(20261003)
    seeds = np.array([0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144])
    done = 0
    for _ in range(160):
        ts, n_inds, out = _random_multigraph(rng)
        if not _supported(ts, n_inds):
            continue
        om = ct.pack_masks([out], n_inds)[0]
        host = core.greedy_trees(ts, n_inds, seeds, output_mask=om)
        dev = core.greedy_trees(ts, n_inds, seeds, output_mask=om, device=0)
        assert np.array_equal(host, dev), (len(ts), n_inds)
        done += 1
    assert done > 100


def test_both_shuffle_kernels_continue_a_generator(monkeypatch):
    """`draws` (outputs of Random(seed) already consumed) through the LDS kernel and the in-memory one."""
    a = syn.regular_problem(40, graph_seed=2)
    seeds = np.array([3, 99, 2**31 + 5, 7, 8, 9, 2**32 - 1])
    got = {}
    for lds in ("1", "0"):
        monkeypatch.setenv("TNCO_HIP_SHUFFLE_LDS", lds)
        draws = np.array([0, 1, 623, 624, 625, 5000, 12], np.uint64)
        got[lds] = (_both(a.ts_inds, a.n_inds, seeds, draws=draws), draws.copy())
    assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1])


def test_config5_topology_and_many_seeds(form):
    p = syn.sycamore_problem(20)
    assert _supported(p.ts_inds, p.n_inds)
    _both(p.ts_inds, p.n_inds, syn.replica_seeds(600))


def test_components_share_one_generator():
    a, b = syn.regular_problem(10, graph_seed=2), syn.regular_problem(16, graph_seed=4)
    seeds = np.array([3, 99, 2**31 + 5, 7, 8, 9])
    draws = np.zeros(len(seeds), np.uint64)
    _both(a.ts_inds, a.n_inds, seeds, draws=draws)
    assert np.all(draws >= 9)
    _both(b.ts_inds, b.n_inds, seeds, draws=draws)


def test_networks_outside_the_kernel_limits_take_the_host_path():
    # an index held by seven tensors: not the kernel's, same result through the same entry point
    ts = [[0, i + 1, (i + 1) % 7 + 1] for i in range(7)]
    assert not _supported(ts, 8)
    _both(ts, 8, syn.replica_seeds(5), on_device=False)
    # two tensors / one tensor: contract_path does not call the optimizer
    _both([[0, 1], [1, 2]], 3, [1, 2, 3], on_device=False)


def test_trees_stay_on_the_device_for_the_optimizer():
    """greedy_trees(keep_on_device=True) -> BatchedOptimizer without a trip through the host: the same
    optimizer state as from the host array; tnco_hip_create checks such trees on the device."""
    prob = syn.regular_problem(64, graph_seed=7)
    seeds = syn.replica_seeds(300)
    host = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
    dl = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0, keep_on_device=True)
    assert isinstance(dl, core.DeviceLinks) and dl.shape == host.shape
    assert np.array_equal(dl.numpy(), host)
    a = core.BatchedOptimizer(prob.leaf_masks, dl, seeds, n_inds=prob.n_inds)
    b = core.BatchedOptimizer(prob.leaf_masks, host, seeds, n_inds=prob.n_inds)
    betas = syn.linear_betas(0, 50, 40)
    a.run(betas)
    b.run(betas)
    assert np.array_equal(a.costs()[0], b.costs()[0]) and np.array_equal(a.costs()[1], b.costs()[1])
    for r in (0, 17, 299):
        assert all(np.array_equal(x, y) for x, y in zip(a.tree(r), b.tree(r)))
    # a network outside the kernel's limits: the host version's trees, uploaded
    ts = [[0, i + 1, (i + 1) % 7 + 1] for i in range(7)]
    dl2 = core.greedy_trees(ts, 8, seeds[:5], device=0, keep_on_device=True)
    assert np.array_equal(dl2.numpy(), core.greedy_trees(ts, 8, seeds[:5]))


def test_create_rejects_bad_trees_in_device_memory():
    """The device-side twin of the host check (Node::is_valid / Tree::is_valid, node.hpp:72-107,
    tree.hpp:58-139): same messages."""
    prob = syn.regular_problem(16, graph_seed=3)
    seeds = syn.replica_seeds(4)
    dl = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0, keep_on_device=True)
    good = dl.numpy()
    import torch
    cases = [(2, 2, 30, 5), (1, 0, 20, 3), (3, 0, 2, 1), (0, 2, 4, 4), (2, 1, 25, -1), (1, 2, 7, 29)]
    seen = set()
    for r, row, col, val in cases:
        bad = good.copy()
        bad[r, row, col] = val
        t = torch.from_numpy(bad).cuda()
        with pytest.raises(ValueError) as on_host:
            core.BatchedOptimizer(prob.leaf_masks, bad, seeds, n_inds=prob.n_inds)
        with pytest.raises(ValueError) as on_device:
            core.BatchedOptimizer(prob.leaf_masks, core.DeviceLinks(t.data_ptr(), bad.shape, 0), seeds, n_inds=prob.n_inds)
        assert str(on_device.value) == str(on_host.value)
        seen.add(str(on_host.value))
    assert len(seen) >= 3  # several of the checks were exercised
