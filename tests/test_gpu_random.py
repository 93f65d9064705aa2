"""Randomised GPU parity (hypothesis): random small networks with random combinations of the
options of the path -- hyper-indices, output legs, uniform / per-index dims, sparse legs, cost type,
acceptance rule, disable_shared_inds, finite width with random bounds -- bit-exact against the
oracle, every replica valid on the device."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, example, given, settings, strategies as st

from tests import helpers as H
from tnco_amd import synthetic as syn

pytestmark = pytest.mark.gpu

# fixed examples by default (a red round-end run must be reproducible); TNCO_FUZZ=N draws N fresh ones
_N = int(os.environ.get("TNCO_FUZZ", "0"))
SET = dict(max_examples=_N or 14, derandomize=not _N, deadline=None,
           suppress_health_check=[HealthCheck.function_scoped_fixture])


@pytest.fixture(scope="module")
def core():
    from tnco_amd import core as c
    return c


def _expected_validate(verdicts):
    bad = [r for r, v in enumerate(verdicts) if v]
    return (len(bad), bad[0] if bad else -1)


def _problem(seed, n, k, dims_kind, n_sparse):
    ts, dims, out = syn.random_hyper_tn(n, int(2.2 * n), k=k, n_output=seed % 4, seed=seed,
                                        dims_choices=((2, 3, 4) if seed % 2 else (2, 3, 5, 6, 8)) if dims_kind == "vector" else (2,))
    n_inds = 1 + max(i for xs in ts for i in xs)
    d = {"two": 2, "three": 3, "four": 4, "vector": np.array(dims[:n_inds], np.uint64)}[dims_kind]
    rng = np.random.RandomState(seed)
    sparse = sorted(int(x) for x in rng.choice(n_inds, size=min(n_sparse, n_inds), replace=False)) if n_sparse else ()
    return H.Problem(ts, d, out, sparse_inds=sparse)


@settings(**SET)
@given(seed=st.integers(0, 10**6), n=st.integers(4, 40), k=st.sampled_from([2, 3, 4]),
       dims_kind=st.sampled_from(["two", "three", "four", "vector"]), n_sparse=st.sampled_from([0, 0, 3, 8]),
       cost_type=st.sampled_from(["float64", "float64", "float32"]), kind=st.sampled_from(["mh", "mh", "greedy", "base"]),
       dsi=st.booleans())
def test_random_infinite_memory(core, oracle_lib, seed, n, k, dims_kind, n_sparse, cost_type, kind, dsi):
    prob = _problem(seed, n, k, dims_kind, n_sparse)
    seeds = H.replica_seeds(6, S=seed)
    links = prob.links(seeds)
    kw = dict(cost_type=cost_type, disable_shared_inds=dsi)
    if n_sparse:
        kw["n_projs"] = 1 + seed % 9
    try:
        gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=prob.dims,
                                    output_mask=prob.output_mask, sparse_mask=prob.sparse_mask, **kw)
    except ValueError as e:  # float32 overflow of a random tree: the oracle must refuse it too
        assert "Precision is too low" in str(e)
        with pytest.raises(ValueError, match="Precision is too low"):
            for r in range(len(seeds)):
                H.make_oracle(oracle_lib, prob, links[r], seeds[r], **kw)
        return
    betas = H.linear_betas(0, 10 + seed % 60, 40 + seed % 50)
    gpu.run(betas[:17], kind)
    gpu.run(betas[17:], kind)
    tot, mn = gpu.costs()
    verdicts = []
    for r in range(len(seeds)):
        o = H.make_oracle(oracle_lib, prob, links[r], seeds[r], **kw)
        o.run({"base": 0, "greedy": 1, "mh": 2}[kind], betas)
        H.assert_replica_equal(gpu, r, o)
        assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        verdicts.append(o.is_valid() != 0)
    # is_valid() per replica: the same verdicts as the oracle's (a float32 cost that overflowed to inf
    # during the run is "not valid" for the reference too: log(inf) - log(inf) is not <= atol)
    assert gpu.validate() == _expected_validate(verdicts)
    gpu.close()


@settings(**SET)
@given(seed=st.integers(0, 10**6), n=st.integers(6, 36), k=st.sampled_from([2, 3]),
       dims_kind=st.sampled_from(["two", "two", "four", "vector"]), n_sparse=st.sampled_from([0, 0, 4]),
       frac=st.floats(0.3, 1.1), every=st.sampled_from([1, 3, 10]), width_type=st.sampled_from(["float32", "float64"]),
       new_slices=st.sampled_from([0, 0, 2]))
# (large networks, found by tools/fuzz_gpu.py --nmin 150 --nmax 700: 12- and 15-word masks, trees deeper
# than the LDS traversal stack, the max_number_new_slices branch with its full rebuilds)
@example(seed=717032, n=501, k=3, dims_kind="two", n_sparse=0, frac=0.3146844055637413, every=3,
         width_type="float32", new_slices=2)
@example(seed=368628, n=315, k=3, dims_kind="two", n_sparse=0, frac=0.7781581243643569, every=10,
         width_type="float64", new_slices=0)
def test_random_finite_width(core, oracle_lib, seed, n, k, dims_kind, n_sparse, frac, every, width_type, new_slices):
    prob = _problem(seed, n, k, dims_kind, n_sparse)
    seeds = H.replica_seeds(5, S=seed)
    links = prob.links(seeds)
    # a bound relative to the widest tensor of the first initial tree (in log2 units)
    l, r, _p = links[0]
    from tnco_amd import ctree as ct
    dimlist = [int(x) for x in (prob.dims if np.ndim(prob.dims) else [prob.dims] * prob.n_inds)]
    w0 = max(sum(np.log2(dimlist[x]) for x in ct.unpack_mask(m)) for m in prob.node_masks(l, r))
    max_width = max(1.0, float(np.float32(frac * w0)))
    kw = dict(width_type=width_type, max_number_new_slices=new_slices)
    if n_sparse:
        kw["n_projs"] = 2 + seed % 7
    gpu = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=prob.dims,
                                output_mask=prob.output_mask, sparse_mask=prob.sparse_mask, max_width=max_width, **kw)
    betas = H.linear_betas(0, 40, 30 + seed % 30)
    gpu.run(betas[:11], "mh", update_slices_every=every)
    gpu.run(betas[11:], "mh", update_slices_every=every)
    tot, mn = gpu.costs()
    verdicts = []
    for r in range(len(seeds)):
        o = H.make_oracle(oracle_lib, prob, links[r], seeds[r], max_width=max_width, **kw)
        o.run(oracle_lib.PROB_MH, betas, update_slices_every=every)
        H.assert_replica_equal(gpu, r, o)
        gs, gms = gpu.slices(r)
        os_, oms = o.slices()
        assert np.array_equal(gs, os_) and np.array_equal(gms, oms)
        assert tot[r] == o.total_cost and mn[r] == o.min_total_cost
        verdicts.append(o.is_valid() != 0)
    # same verdicts as the oracle: with sparse legs the max_number_new_slices branch tracks the sliced
    # width by subtracting log2(dims) per new slice (greedy/optimizer.hpp:262-268), not through the
    # sparse width model, so the reference itself can accept a tensor its is_valid() then rejects
    assert gpu.validate() == _expected_validate(verdicts)
    gpu.close()
