"""Recipes of the golden SA cases (tests/golden/sa_golden.json) and the state digest both the
generator script and the tests use."""
import numpy as np

from tests import helpers as H
from tnco_amd import synthetic as syn

# name, network recipe, replicas, schedule; finite width: max_width / update_slices
CASES = [
    dict(name="G1 README chain, 4 tensors (fuse off)", tn=("chain", 4), n_replicas=4, seed_base=1, betas=(0, 20, 20), every=5),
    dict(name="G2 config-2 topology, 64-leaf 3-regular", tn=("regular", 64, 7), n_replicas=8, seed_base=2,
         betas=(0, 100, 300), every=100),
    dict(name="G3 config-3 topology, 512-leaf 3-regular", tn=("regular", 512, 11), n_replicas=2, seed_base=3,
         betas=(0, 100, 200), every=100),
    dict(name="G4a per-index dims {2,3,4,6} with hyper-indices and outputs", tn=("hyper", 28, 60, 3, 4, (2, 3, 4, 6)),
         n_replicas=4, seed_base=4, betas=(0, 40, 120), every=60),
    dict(name="G4b sparse legs, float32 cost", tn=("hyper", 24, 50, 2, 0, (2,)), sparse=(1, 5, 9, 20, 33), n_projs=6,
         cost_type="float32", n_replicas=4, seed_base=5, betas=(0, 40, 120), every=60),
    dict(name="G5 finite width, 64-leaf 3-regular, max_width 7", tn=("regular", 64, 3), max_width=7.0, update_slices=10,
         n_replicas=6, seed_base=6, betas=(0, 80, 200), every=50),
    dict(name="G5b finite width on the config-5 topology (Sycamore-53 style, depth 20), max_width 40", tn=("sycamore", 20),
         max_width=40.0, update_slices=10, n_replicas=2, seed_base=7, betas=(0, 100, 60), every=30),
    dict(name="G6 greedy acceptance", tn=("regular", 40, 9), prob=1, n_replicas=4, seed_base=8, betas=(0, 0, 60), every=30),
]


def problem_of(case):
    kind = case["tn"][0]
    if kind == "chain":
        ts, d, out = syn.chain_tn(case["tn"][1])
        prob = H.Problem(ts, d, out)
    elif kind == "regular":
        prob = H.regular_problem(case["tn"][1], graph_seed=case["tn"][2])
    elif kind == "sycamore":
        prob = syn.sycamore_problem(case["tn"][1], "alternating")  # (the easier network of rounds 1-3: the fixture was made on it)
    else:
        _k, n, n_inds, k, n_out, choices = case["tn"]
        ts, dims, out = syn.random_hyper_tn(n, n_inds, k=k, n_output=n_out, seed=case["seed_base"], dims_choices=choices)
        d = dims[0] if len(set(dims)) == 1 else np.array(dims, np.uint64)
        prob = H.Problem(ts, d, out, sparse_inds=case.get("sparse", ()))
    seeds = H.replica_seeds(case["n_replicas"], S=case["seed_base"])
    links = prob.links(seeds)
    b0, b1, n = case["betas"]
    betas = H.linear_betas(b0, b1, n) if b0 != b1 else np.full(n, float(b0))
    okw = {k: case[k] for k in ("cost_type", "n_projs", "max_width") if k in case}
    return prob, seeds, links, betas, okw, case["every"]


def _fnv(h, a):
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def state_hash(cur, best, total, mn, prng, slices):
    h = 1469598103934665603
    for t in (cur, best):
        for a in t:
            h = _fnv(h, a)
    rec = {"trees": f"{h:016x}", "total": float(total).hex(), "min": float(mn).hex(), "prng_pos": int(prng[624]),
           "prng": f"{_fnv(1469598103934665603, prng):016x}"}
    if slices is not None:
        rec["slices"] = [f"{_fnv(1469598103934665603, s):016x}" for s in slices]
    return rec
