"""The pin that turns parity green the day the image has Boost: the oracle's
Optimizer::update / cost models / get_slices against the REAL reference module, sweep by sweep.

`make -C oracle ref_core` compiles /root/reference/include/tnco/main.cpp where it lies into oracle/_ref/ -- ONLY when a
genuine <boost/dynamic_bitset.hpp> is on the compiler's system include path (oracle/Makefile; this image has none,
and no stand-in header is written: a module built against one pins nothing).  Without that module every test here is
SKIPPED, and DESIGN.md section 5 keeps saying "parity unpinned by reference execution".

What is compared after every chunk of sweeps, bit for bit: the links of ctree and min_ctree, every leg mask, the text of
the mt19937 state, log2 of total_cost / min_total_cost (doubles), slices / min_slices -- on the cases of SURVEY 8(c)
G1-G5: the BaseOptimization example and the README chain, the 64- and 512-leaf 3-regular networks, per-index dims,
sparse legs, float32 cost, hyper-indices, all three probabilities, and the finite-width optimizer with re-slicing.
"""
import importlib.util
import math
import os
import sys
from pathlib import Path

import numpy as np
import pytest

from tests import helpers as H

ROOT = Path(__file__).resolve().parent.parent
REF_DIR = Path(os.environ.get("TNCO_REF_CORE_DIR", ROOT / "oracle" / "_ref"))


def _load_core():
    so = sorted(REF_DIR.glob("tnco_core*.so"))
    if not so:
        return None
    spec = importlib.util.spec_from_file_location("tnco_core", so[0])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["tnco_core"] = mod  # (its submodules register under this name)
    spec.loader.exec_module(mod)
    return mod


core = _load_core()
pytestmark = pytest.mark.skipif(core is None, reason="no oracle/_ref/tnco_core*.so: the reference needs Boost, absent in this "
                                                     "image (make -C oracle ref_core builds it where Boost exists)")


def _bits(mask_row, n_inds):
    return [p for p in range(n_inds) if (int(mask_row[p >> 6]) >> (p & 63)) & 1]


def _ref_ctree(prob, links, check_shared=True):
    l, r, p = links
    masks = prob.node_masks(l, r)
    def node(i):  # (node.hpp:121-127: a leaf has a parent only, the root children only)
        if l[i] < 0:
            return core.Node(parent=int(p[i]))
        if p[i] < 0:
            return core.Node(children=(int(l[i]), int(r[i])))
        return core.Node((int(l[i]), int(r[i])), int(p[i]))

    nodes = [node(i) for i in range(len(l))]
    inds = [core.Bitset(_bits(masks[i], prob.n_inds), prob.n_inds) for i in range(len(l))]
    dims = int(prob.dims) if np.ndim(prob.dims) == 0 else [int(d) for d in prob.dims]
    return core.ContractionTree(nodes, inds, dims, check_shared_inds=check_shared)


def _ref_state(ct, n_inds):
    W = max(1, (n_inds + 63) // 64)
    N = len(ct.nodes)
    l, r, p = (np.full(N, -1, np.int32) for _ in range(3))
    m = np.zeros((N, W), np.uint64)
    for i, nd in enumerate(ct.nodes):
        c = nd.children
        if c[0] is not None:
            l[i], r[i] = c[0], c[1]
        if nd.parent is not None:
            p[i] = nd.parent
        for q in ct.inds[i].positions():
            m[i, q >> 6] |= np.uint64(1) << np.uint64(q & 63)
    return l, r, p, m


def _mask_of(bitset, W):
    m = np.zeros(W, np.uint64)
    for q in bitset.positions():
        m[q >> 6] |= np.uint64(1) << np.uint64(q & 63)
    return m


def _compare(ref, orc_o, n_inds, fw=False, f32=False):
    for which in (False, True):
        rl, rr, rp, rm = _ref_state(ref.min_ctree if which else ref.ctree, n_inds)
        ol, orr, op, om = orc_o.tree(which_min=which)
        assert np.array_equal(rl, ol) and np.array_equal(rr, orr) and np.array_equal(rp, op), f"links (min={which})"
        assert np.array_equal(rm, om), f"leg masks (min={which})"
    assert ref.prng_state.split() == [str(int(x)) for x in orc_o.prng_state()], "mt19937 state"
    if f32:  # (the module returns log2f of a float: the cost's bits cannot be recovered from it -- two float ulps; a cost
        # that differed would change a decision, and with it the trees and the generator compared above)
        assert abs(ref.log2_total_cost - math.log2(orc_o.total_cost)) <= 2.4e-7 * abs(ref.log2_total_cost), "total_cost"
        assert abs(ref.log2_min_total_cost - math.log2(orc_o.min_total_cost)) <= 2.4e-7 * abs(ref.log2_min_total_cost), "min_total_cost"
    else:
        assert ref.log2_total_cost == math.log2(orc_o.total_cost), "total_cost"
        assert ref.log2_min_total_cost == math.log2(orc_o.min_total_cost), "min_total_cost"
    if fw:
        s, ms = orc_o.slices()
        assert np.array_equal(_mask_of(ref.slices, len(s)), s) and np.array_equal(_mask_of(ref.min_slices, len(ms)), ms), "slices"
    assert ref.is_valid() and orc_o.is_valid() == 0


def _lockstep(oracle_lib, prob, links, seed, betas, *, kind="mh", cost="float64", chunk=10, max_width=None, width="float32",
              every=10, n_projs=None, disable_shared=False):
    om, pr = core.optimize, core.optimize.prob
    okw = dict(cost_type=cost, disable_shared_inds=disable_shared, n_projs=n_projs or 0)
    if max_width is None:
        cm = om.infinite_memory.cost_model
        cmodel = (getattr(cm, f"SimpleCostModelSparseInds_{cost}")(core.Bitset(_bits(prob.sparse_mask, prob.n_inds), prob.n_inds), n_projs)
                  if prob.sparse_mask is not None else getattr(cm, f"SimpleCostModel_{cost}")())
        ref = getattr(om.infinite_memory, f"Optimizer_{cost}")(_ref_ctree(prob, links, not disable_shared), cmodel, seed=int(seed),
                                                                disable_shared_inds=disable_shared)
    else:
        cm = om.finite_width.cost_model
        cmodel = getattr(cm, f"SimpleCostModel_{cost}_{width}")(float(max_width))
        ref = getattr(om.finite_width.greedy, f"Optimizer_{cost}_{width}")(_ref_ctree(prob, links, not disable_shared), cmodel,
                                                                            seed=int(seed), disable_shared_inds=disable_shared)
        okw.update(max_width=max_width, width_type=width)
    o = H.make_oracle(oracle_lib, prob, links, seed, **okw)
    P = {"base": getattr(pr, f"BaseProbability_{cost}")(), "greedy": getattr(pr, f"Greedy_{cost}")(),
         "mh": getattr(pr, f"MetropolisHastings_{cost}")(0.0)}[kind]
    k = {"base": oracle_lib.PROB_BASE, "greedy": oracle_lib.PROB_GREEDY, "mh": oracle_lib.PROB_MH}[kind]
    _compare(ref, o, prob.n_inds, max_width is not None, cost == "float32")  # the constructors: caches, min cost, (finite width) first get_slices
    for lo in range(0, len(betas), chunk):
        for j, b in enumerate(betas[lo:lo + chunk]):
            if kind == "mh":
                P.beta = float(b)
            if max_width is None:
                ref.update(P)
            else:  # tnco/app/finite_width/sa.py:228
                ref.update(P, update_slices=((lo + j) % every == 0))
        o.run(k, betas[lo:lo + chunk], **({} if max_width is None else {"update_slices_every": every}))
        _compare(ref, o, prob.n_inds, max_width is not None, cost == "float32")


def test_g1_notebook_example_and_readme_chain(oracle_lib):
    from tnco_amd.synthetic import Problem
    chain = Problem([[0], [0, 1], [1, 2], [2]], 2)  # README: "2 a b / 2 b c / 2 c d", un-fused
    for seed in range(6):
        links = chain.links([seed])[0]
        for kind in ("mh", "greedy", "base"):
            _lockstep(oracle_lib, chain, links, seed, H.linear_betas(0, 10, 20), kind=kind, chunk=1)


@pytest.mark.parametrize("n,graph_seed,sweeps,n_seeds", [(64, 7, 400, 8), (512, 11, 120, 2)])
def test_g2_g3_regular_networks(oracle_lib, n, graph_seed, sweeps, n_seeds):
    prob = H.regular_problem(n, graph_seed=graph_seed)
    seeds = H.replica_seeds(n_seeds)
    links = prob.links(seeds)
    for r, s in enumerate(seeds):
        _lockstep(oracle_lib, prob, links[r], s, H.linear_betas(0, 100, sweeps), chunk=20)


def test_g4_cost_models(oracle_lib):
    from tnco_amd import synthetic as syn
    ts, dims, out = syn.random_hyper_tn(28, 70, k=3, n_output=3, seed=8, dims_choices=(2, 3, 5, 2, 7))
    prob = H.Problem(ts, np.array(dims, np.uint64), out)
    seeds = H.replica_seeds(4, S=8)
    links = prob.links(seeds)
    for r, s in enumerate(seeds):
        _lockstep(oracle_lib, prob, links[r], s, H.linear_betas(0, 30, 100))                      # per-index dims + hyper-indices
        _lockstep(oracle_lib, prob, links[r], s, H.linear_betas(0, 30, 100), cost="float32")
        _lockstep(oracle_lib, prob, links[r], s, H.linear_betas(0, 30, 60), disable_shared=True)
    sp = H.Problem(ts, np.array(dims, np.uint64), out, sparse_inds=[2, 3, 11, 30, 31, 60])
    for r, s in enumerate(seeds[:2]):
        _lockstep(oracle_lib, sp, links[r], s, H.linear_betas(0, 30, 100), n_projs=6)


@pytest.mark.parametrize("width", ["float32", "float64"])
def test_g5_finite_width(oracle_lib, width):
    prob = H.regular_problem(64, graph_seed=7)
    seeds = H.replica_seeds(4, S=3)
    links = prob.links(seeds)
    for r, s in enumerate(seeds):
        o = H.make_oracle(oracle_lib, prob, links[r], s)
        w0 = max(bin(int(x)).count("1") for row in o.tree()[3] for x in [int.from_bytes(row.tobytes(), "little")])
        _lockstep(oracle_lib, prob, links[r], s, H.linear_betas(0, 60, 150), max_width=max(2, int(0.6 * w0)), width=width, chunk=10)
