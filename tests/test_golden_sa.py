"""Golden state vectors of the SA path (tests/golden/sa_golden.json, written by
tests/golden/make_sa_golden.py from the CPU oracle): the oracle still produces them (CPU), and the
HIP path produces them through the C ABI (GPU) -- trees, legs, costs, slices and the PRNG stream at
several points of every schedule, for the README chain, the config-2 / 3 / 5 topologies, per-index
dims with hyper-indices, sparse legs in float32, finite width and the greedy rule."""
import json
from pathlib import Path

import numpy as np
import pytest

from tests import helpers as H
from tests.golden_cases import problem_of, state_hash

GOLD = json.loads((Path(__file__).parent / "golden" / "sa_golden.json").read_text())
IDS = [g["case"]["name"].split(" ")[0] for g in GOLD]


def _case(g):
    case = dict(g["case"])
    case["tn"] = tuple(tuple(x) if isinstance(x, list) else x for x in case["tn"])
    case["betas"] = tuple(case["betas"])
    return case


@pytest.mark.parametrize("g", GOLD, ids=IDS)
def test_oracle_reproduces_the_golden_states(oracle_lib, g):
    case = _case(g)
    prob, seeds, links, betas, okw, every = problem_of(case)
    fw = "max_width" in case
    for r, s in enumerate(seeds):
        o = H.make_oracle(oracle_lib, prob, links[r], s, **okw)
        for k, k0 in enumerate(range(0, len(betas), every)):
            o.run(case.get("prob", 2), betas[k0:k0 + every], **({"update_slices_every": case["update_slices"]} if fw else {}))
            got = state_hash(o.tree(), o.tree(which_min=True), o.total_cost, o.min_total_cost, o.prng_state(),
                             o.slices() if fw else None)
            assert got == g["replicas"][r][k], f"{case['name']}: replica {r}, after {k0 + every} sweeps"


@pytest.mark.gpu
@pytest.mark.parametrize("g", GOLD, ids=IDS)
def test_gpu_reproduces_the_golden_states(g):
    from tnco_amd import core
    case = _case(g)
    prob, seeds, links, betas, okw, every = problem_of(case)
    fw = "max_width" in case
    kw = dict(okw)
    if prob.sparse_mask is not None:
        kw["sparse_mask"] = prob.sparse_mask
    with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=prob.dims,
                               output_mask=prob.output_mask, **kw) as gpu:
        kind = {0: "base", 1: "greedy", 2: "mh"}[case.get("prob", 2)]
        for k, k0 in enumerate(range(0, len(betas), every)):
            gpu.run(betas[k0:k0 + every], kind, update_slices_every=case.get("update_slices", 10))
            tot, mn = gpu.costs()
            for r in range(len(seeds)):
                got = state_hash(gpu.tree(r), gpu.tree(r, which_min=True), tot[r], mn[r], gpu.prng_state(r),
                                 gpu.slices(r) if fw else None)
                assert got == g["replicas"][r][k], f"{case['name']}: replica {r}, after {k0 + every} sweeps"
        assert gpu.validate() == (0, -1)
