"""The worked examples of the reference's own docstrings (tests/golden/reference_doc_examples.json,
collected by tests/golden/make_reference_doc_examples.py from /root/reference) run against this
build's restatements of the same interfaces, doctest style: the statements of an example are executed
with the reference's names bound to the build's functions, and every checked expression must print
what the reference's docstring says.  These are the only worked input / output pairs the reference
holds for the hot path's host side; the optimizer's update() itself has none
(parity unpinned by reference execution)."""
import json
from pathlib import Path

import numpy as np
import pytest

from tnco_amd import ctree as ct
from tnco_amd.app import Optimizer, load_tn
from tnco_amd.app import tn as atn
from tnco_amd.app._sa_driver import merge_contraction_paths, split_contraction_path

EXAMPLES = json.loads((Path(__file__).parent / "golden" / "reference_doc_examples.json").read_text())


class Bitset:
    """tnco/bitset.py:33-117 as this build stores it: bit p of word p // 64 <-> position p; the string
    form is position-0-first (bitset.py:40-46)."""

    def __init__(self, positions, n):
        self.n = n
        self.words = ct.pack_masks([list(positions)], n)[0]

    def __str__(self):
        pos = set(ct.unpack_mask(self.words))
        return "".join("1" if p in pos else "0" for p in range(self.n))


def _prob(kind):
    """tnco/optimize/prob.py over include/tnco/optimize/prob/{base,greedy,mh}.hpp: the oracle's restatement."""
    from oracle import oracle as orc
    orc.build()

    class P:
        def __init__(self, beta=0.0, cost_type="float64"):
            self.beta = beta

        def __call__(self, delta, old):
            return orc.prob(kind, self.beta, float(delta), float(old))
    return P


def get_random_contraction_path(ts_inds, output_inds, seed=None):
    names = list(dict.fromkeys(i for xs in ts_inds for i in xs))
    imap = {x: k for k, x in enumerate(names)}
    cnt = {x: sum(x in xs for xs in ts_inds) for x in names}
    con = ct.greedy_contraction([[imap[i] for i in xs] for xs in ts_inds],
                                [imap[x] for x in output_inds if cnt.get(x, 0) <= 1], seed)
    return [tuple(sorted(p)) for p in ct.ssa_to_linear(con, len(ts_inds))]


def contract(path, ts_inds, arrays=None):
    inds, output = atn.contract(path, ts_inds)
    return inds, output, None  # (numeric arrays are outside the hot path: SURVEY.md section 2)


NAMESPACE = {
    "np": np, "ContractionTree": ct.ContractionTree, "merge_contraction_paths": merge_contraction_paths,
    "split_contraction_path": split_contraction_path, "fuse": atn.fuse, "contract": contract,
    "get_random_contraction_path": get_random_contraction_path, "Bitset": Bitset, "load_tn": load_tn,
    "Optimizer": Optimizer,
}
OUT_OF_SCOPE = {"res[0]"}  # the numeric result of contracting arrays


@pytest.mark.parametrize("ex", EXAMPLES, ids=[e["source"] for e in EXAMPLES])
def test_reference_docstring_example(ex):
    ns = dict(NAMESPACE)
    if "prob.py" in ex["source"]:
        ns.update(BaseProbability=_prob(0), Greedy=_prob(1), MetropolisHastings=_prob(2))
    checks = {c["expr"]: c for c in ex["checks"]}
    # the statements in source order: set-up statements are executed, checked expressions evaluated
    order = sorted([(None, s) for s in ex["setup"]] + [(c["line"], c["expr"]) for c in ex["checks"]],
                   key=lambda t: (t[0] is not None, t[0] or 0))
    for stmt in ex["setup"]:
        exec(stmt, ns)
    for _line, expr in [t for t in order if t[0] is not None]:
        if expr in OUT_OF_SCOPE:
            continue
        got = eval(expr, ns)
        want = checks[expr]["expected"]
        assert repr(got) == want or str(got) == want, f"{ex['source']}: {expr} -> {got!r}, reference says {want}"


def test_examples_cover_the_expected_interfaces():
    srcs = " ".join(e["source"] for e in EXAMPLES)
    for name in ("ctree.py", "utils/tn.py", "bitset.py", "optimize/prob.py", "app/app.py"):
        assert name in srcs
    assert sum(len(e["checks"]) for e in EXAMPLES) >= 12


def test_split_is_the_inverse_of_merge():
    paths = [[(0, 1), (1, 4)], [(4, 5), (2, 4)]]
    merged = merge_contraction_paths(6, paths, autocomplete=False)
    assert split_contraction_path(6, merged) == [[tuple(p) for p in q] for q in paths]
