"""The reference-side binding shown in INTEGRATION.md (`tnco/app/infinite_memory/sa_hip.py`) is
extracted from the document and executed against libtnco_hip.so.

The reference's Python cannot be imported in this image (more_itertools / opt_einsum / autoray are
absent), so the stub gets a stand-in `tnco` namespace that exposes the few reference interfaces it
uses -- `BaseOptimizer`, `ContractionResults`, `ContractionTree(path, ts_inds, dims, output_inds=,
check_shared_inds=)` with `.nodes[i].children / .parent`, `.inds[i].positions()`, `.dims`,
`._inds_order`, `._tensors_pos`, `get_random_contraction_path(..., merge_paths=False, seed=)`,
`merge_contraction_paths` -- backed by this repository's host code.  CPU: the stub loads the library
and binds every entry point it names.  GPU: `Optimizer(method='sa_hip')` of the stub returns what
`tnco_amd.app.Optimizer(method='sa')` returns."""
import re
import sys
import types
from pathlib import Path
from random import Random

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _stub_source():
    md = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    src = [b for b in blocks if "sa_hip.py" in b.split("\n")[0]]
    assert len(src) == 1
    assert "..." not in src[0]  # nothing elided
    return src[0]


class _Node:
    def __init__(self, children, parent):
        self.children, self.parent = children, parent


class _Bits:
    def __init__(self, pos):
        self._pos = list(pos)

    def positions(self):
        return self._pos


def _fake_tnco(monkeypatch):
    from tnco_amd import ctree as ct
    from tnco_amd.app import app as aapp
    from tnco_amd.app import tn as atn
    from tnco_amd.app.infinite_memory import sa as asa

    class ContractionTree:
        def __init__(self, path, ts_inds, dims, *, output_inds=None, check_shared_inds=False):
            t = ct.ContractionTree(path, ts_inds, dims, output_inds=output_inds, check_shared_inds=check_shared_inds)
            self.nodes = [_Node((int(l), int(r)), int(p)) for l, r, p in zip(t.left, t.right, t.parent)]
            self.inds = [_Bits(ct.unpack_mask(m)) for m in t.masks]
            self.n_leaves = t.n_leaves
            self.dims = t.dims
            self._inds_order = t.inds_order
            self._tensors_pos = t.tensors_pos

    def get_random_contraction_path(ts_inds, output_inds, *, merge_paths=True, seed=None):
        """tnco/utils/tn.py:109-273 with merge_paths=False: one linear path per connected component."""
        assert merge_paths is False
        ts_inds = [list(x) for x in ts_inds]
        holders = {}
        for xs in ts_inds:
            for i in xs:
                holders[i] = holders.get(i, 0) + 1
        rng = Random(seed)
        paths = []
        for cc in atn.get_connected_components(ts_inds):
            if len(cc) <= 1:
                paths.append([])
                continue
            order = tuple(dict.fromkeys(i for t in cc for i in ts_inds[t]))
            imap = {x: k for k, x in enumerate(order)}
            keep = [imap[x] for x in output_inds if x in imap and holders[x] <= 1]
            con = ct.greedy_contraction([[imap[i] for i in ts_inds[t]] for t in cc], keep, seed, rng=rng)
            nc, shift = len(cc), len(ts_inds) - len(cc)
            resc = [tuple(cc[p] if p < nc else p + shift for p in xs) for xs in con]
            paths.append([tuple(sorted(p)) for p in ct.ssa_to_linear(resc, len(ts_inds))])
        return paths

    mods = {name: types.ModuleType(name) for name in
            ("tnco", "tnco.app", "tnco.app.app", "tnco.app.infinite_memory", "tnco.app.infinite_memory.sa",
             "tnco.ctree", "tnco.utils", "tnco.utils.tn")}
    mods["tnco.app.app"].BaseOptimizer = aapp.BaseOptimizer
    mods["tnco.app.infinite_memory.sa"].ContractionResults = asa.ContractionResults
    mods["tnco.ctree"].ContractionTree = ContractionTree
    mods["tnco.utils.tn"].get_random_contraction_path = get_random_contraction_path
    mods["tnco.utils.tn"].merge_contraction_paths = asa.merge_contraction_paths
    mods["tnco.utils"].tn = mods["tnco.utils.tn"]
    mods["tnco"].utils = mods["tnco.utils"]
    for name, m in mods.items():
        monkeypatch.setitem(sys.modules, name, m)
    monkeypatch.setenv("TNCO_HIP_LIB", str(ROOT / "tnco_amd" / "libtnco_hip.so"))


def _load_stub(monkeypatch):
    _fake_tnco(monkeypatch)
    ns = {"__name__": "tnco.app.infinite_memory.sa_hip"}
    exec(compile(_stub_source(), "INTEGRATION.md:sa_hip.py", "exec"), ns)
    return ns


def test_stub_loads_the_library_and_binds_its_entry_points(monkeypatch):
    ns = _load_stub(monkeypatch)
    from tnco_amd import _lib
    assert [f[0] for f in ns["Desc"]._fields_] == [f[0] for f in _lib.Desc._fields_]
    for name in re.findall(r"_lib\.(tnco_hip_\w+)", _stub_source()):
        assert hasattr(ns["_lib"], name)
    opt = ns["Optimizer"](seed=1)
    with pytest.raises(ValueError, match="'n_steps' must be a positive number."):
        opt.optimize("2 a b\n2 b c", betas=(0, 1), n_steps=0, fuse=None)
    w = ns["_words"]([0, 63, 64, 130], 3)
    assert [int(x) for x in w] == [1 | (1 << 63), 1, 4]


@pytest.mark.gpu
def test_stub_results_equal_the_shipped_driver(monkeypatch):
    ns = _load_stub(monkeypatch)
    from tnco_amd import synthetic as syn
    from tnco_amd.app import Optimizer
    ts, _d, _o = syn.random_regular_tn(20, 3, 4)
    spec = [(2, *[f"t{t}" for t in range(20) if k in ts[t]]) for k in range(30)]
    spec += [(3, "u0", "u1", "*"), (3, "u1", "u2"), (2, "lonely")]  # a second component (dims 3, an output leg), a single tensor
    kw = dict(betas=(0, 40), n_steps=120, n_runs=24, fuse=None)
    tn_a, res_a = ns["Optimizer"](seed=7).optimize(spec, **kw)
    tn_b, res_b = Optimizer(method="sa", seed=7).optimize(spec, **kw)
    assert len(res_a) == len(res_b) == 24
    assert [r.cost for r in res_a] == [r.cost for r in res_b]
    key = lambda r: (r.cost, [tuple(map(tuple, p)) for p in r.disconnected_paths])  # noqa: E731
    assert sorted(map(key, res_a)) == sorted(map(key, res_b))
    assert [np.array(r.path).shape for r in res_a] == [(len(tn_a) - 1, 2)] * 24
