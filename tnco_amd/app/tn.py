"""Minimal tensor-network container + loader for the optimizer front door.

Host mirror of the slice of /root/reference/tnco/app/tn.py:76-362 (Tensor,
TensorNetwork) and tnco/app/app.py:154-500 (load_tn) that feeds the SA path:
index-list inputs (list or string form, `*` output token, `/` sparse token;
tnco/utils/tn.py:520-569 read_inds) and the symbolic pre-fusing of tensors
(`fuse`, tnco/utils/tn.py:598-824; app.py:373-414).  Circuits, arrays and
hyper-index decomposition belong to the reference's front-end and are out of
scope.
"""
from __future__ import annotations

import json
import re
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Iterable

__all__ = ["Tensor", "TensorNetwork", "load_tn", "read_inds", "get_connected_components", "get_hyper_count",
           "fuse", "contract"]


@dataclass(frozen=True)
class Tensor:
    inds: tuple
    dims: tuple
    tags: dict = field(default_factory=dict)

    def __post_init__(self):
        inds, dims = tuple(self.inds), tuple(int(d) for d in self.dims)
        problem = ("Wrong number of 'inds'." if len(inds) != len(dims) else
                   "Every dimension must be a positive integers." if min(dims, default=1) < 1 else None)
        if problem:
            raise ValueError(problem)
        self.__dict__.update(inds=inds, dims=dims)  # (a frozen dataclass: normalised in place)

    @property
    def ndim(self) -> int:
        return len(self.dims)

    def to_dict(self):
        return dict(inds=list(self.inds), dims=list(self.dims), array=None, tags=self.tags)


class TensorNetwork:
    """List of tensors + output / sparse indices (tnco/app/tn.py:180-362)."""

    def __init__(self, tensors: Iterable[Tensor], *, output_inds=None, sparse_inds=None, tags=None):
        self.tensors = tuple(tensors)
        dims = {}
        for t in self.tensors:
            for i, d in zip(t.inds, t.dims):
                if dims.setdefault(i, d) != d:
                    raise ValueError("Tensors have indices with different dimensions.")
        self._dims = dims
        count = defaultdict(int)
        for t in self.tensors:
            for i in t.inds:
                count[i] += 1
        if output_inds is None:
            if any(c > 2 for c in count.values()):
                raise ValueError("'output_inds' must be provided if 'TensorNetwork' has hyper-indices.")
            output_inds = [i for i, c in count.items() if c == 1]
        self.output_inds = frozenset(output_inds)
        self.sparse_inds = frozenset(sparse_inds or ())
        if not self.output_inds <= dims.keys() or not self.sparse_inds <= dims.keys():
            raise ValueError("'output_inds' / 'sparse_inds' are not valid.")
        self.tags = dict(tags or {})

    def __len__(self) -> int:
        return len(self.tensors)

    @property
    def ts_inds(self) -> tuple:
        return tuple(t.inds for t in self.tensors)

    @property
    def dims(self) -> dict:
        return dict(self._dims)

    @property
    def n_tensors(self) -> int:
        return len(self.tensors)

    def to_json(self) -> str:
        return json.dumps(dict(tensors=[t.to_dict() for t in self.tensors],
                               output_inds=sorted(self.output_inds, key=str),
                               sparse_inds=sorted(self.sparse_inds, key=str)))


def read_inds(inds_map: dict, *, output_index_token="*", sparse_index_token="/"):
    """The index-list form {index: (dim, tensor, tensor, ...)} turned around into {tensor: (index, ...)}, with the
    dims, the open indices (those listed under the output token) and the sparse ones (contract:
    tnco/utils/tn.py:520-546).  Tensors appear in order of first mention, their indices in the order of the map."""
    if output_index_token == sparse_index_token:
        raise ValueError("'output_index_token' and 'sparse_index_token' must differ.")
    marked = {output_index_token: set(), sparse_index_token: set()}
    dims, held = {}, {}
    for index, entry in inds_map.items():
        entry = tuple(entry)
        dims[index] = int(entry[0])
        for name in entry[1:]:
            if name in marked:
                marked[name].add(index)
            else:
                held.setdefault(name, []).append(index)
    tensor_map = {name: tuple(inds) for name, inds in held.items()}
    return tensor_map, dims, frozenset(marked[output_index_token]), frozenset(marked[sparse_index_token])


def get_connected_components(ts_inds) -> list[tuple[int, ...]]:
    """Connected components as sorted tuples of tensor positions (tnco/utils/tn.py:61-106)."""
    ts_inds = list(ts_inds)
    parent = list(range(len(ts_inds)))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    first = {}
    for t, inds in enumerate(ts_inds):
        for i in inds:
            if i in first:
                ra, rb = find(t), find(first[i])
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
            else:
                first[i] = t
    comps = defaultdict(list)
    for t in range(len(ts_inds)):
        comps[find(t)].append(t)
    return [tuple(v) for v in comps.values()]


def get_hyper_count(ts_inds, output_inds=None) -> dict:
    """index -> (number of tensors holding it) - 1, + 1 for an output index (tnco/utils/tn.py:572-595)."""
    count = {}
    for xs in ts_inds:
        for x in xs:
            count[x] = count.get(x, -1) + 1
    for x in (output_inds or ()):
        count[x] = count.get(x, 0) + 1
    return count


def fuse(ts_inds, dims, max_width, output_inds=None, *, exclude_inds=(), seed=None,
         return_fused_inds: bool = False):
    """Random pairwise pre-contraction of tensors while the result stays within `max_width`
    (log2 of its size); returns the contraction path in linear form.

    Behaviour of tnco/utils/tn.py:598-824, restated: indices that are still to be contracted are
    drawn at random (`Random(seed).randrange` over the list of candidates, in order of first
    appearance); two of the tensors holding the drawn index are drawn (`Random.sample`) and merged
    unless the merged tensor would be wider than `max_width` or would hold an excluded index, in
    which case the index is dropped from the candidates; an index that is still shared after a
    merge (a hyper-index) goes back to the end of the candidate list.  The legs of a merged tensor
    are ordered as they appear in the first, then in the second tensor.

    Parity with the reference is by reading only: tnco.utils.tn cannot be imported in the build
    image (more_itertools, opt_einsum, autoray are absent), so no vectors could be captured.
    """
    import math
    import random
    rng = random.Random(seed)
    live = {t: tuple(xs) for t, xs in enumerate(ts_inds)}
    every = list(dict.fromkeys(x for xs in live.values() for x in xs))  # first-appearance order
    exclude = frozenset(exclude_inds)
    if not exclude <= set(every):
        raise ValueError("'exclude_inds' contains indices not in 'ts_inds'.")
    try:
        dims = dict.fromkeys(every, int(dims))
    except (TypeError, ValueError):
        dims = dict(dims)
    if not set(every) <= dims.keys():
        raise ValueError("'dims' is missing some indices.")
    left = get_hyper_count(live.values())  # contractions each index still takes part in
    if output_inds is None:
        if any(c > 1 for c in left.values()):
            raise ValueError("'output_inds' must be provided if 'ts_inds' has hyper-indices.")
        output_inds = [x for x, c in left.items() if c == 0]
    output = frozenset(output_inds)
    if not output <= set(every):
        raise ValueError("'output_inds' is not consistent with 'ts_inds'.")
    holders = {}
    for t, xs in live.items():
        for x in xs:
            holders.setdefault(x, set()).add(t)
    candidates = [x for x in every if x not in exclude and left[x] != 0]
    next_id = len(live)
    merges = []
    while candidates:
        index = candidates.pop(rng.randrange(len(candidates)))
        if not left.get(index):
            continue
        ta, tb = rng.sample(tuple(holders[index]), k=2)
        xa, xb = live[ta], live[tb]
        sa, sb = frozenset(xa), frozenset(xb)
        if (sa | sb) & exclude:
            continue
        shared = sa & sb
        still_shared = frozenset(x for x in shared if left[x] > 1)
        keep = (sa ^ sb) | still_shared | (output & (sa | sb))
        merged = tuple(dict.fromkeys(x for x in xa + xb if x in keep))
        if sum(math.log2(dims[x]) for x in merged) > max_width:
            continue
        for x in shared:
            left[x] -= 1
        for x in merged:
            holders[x] -= {ta, tb}
            holders[x] |= {next_id}
        for x in shared - still_shared - output:
            del holders[x]
        del live[ta], live[tb]
        live[next_id] = merged
        if left.get(index):
            candidates.append(index)
        merges.append((ta, tb, merged))
        next_id += 1
    # tensor ids -> positions in the shrinking list (every new tensor is appended last)
    slots = list(range(next_id))
    path, fused = [], []
    for ta, tb, merged in merges:
        lo, hi = sorted((ta, tb))
        phi = slots.index(hi)
        del slots[phi]
        plo = slots.index(lo)
        del slots[plo]
        path.append((plo, phi))
        fused.append(merged)
    return (path, fused) if return_fused_inds else path


def contract(path, ts_inds, output_inds=None, dims=None):
    """Symbolic contraction along a linear path: (remaining tensors' indices, remaining output
    indices) -- the `arrays=None` case of tnco/utils/tn.py:906-1072.  An index shared by the two
    tensors survives while other tensors (or the output) still hold it."""
    ts = [tuple(xs) for xs in ts_inds]
    left = get_hyper_count(ts)
    if output_inds is None:
        if any(c > 1 for c in left.values()):
            raise ValueError("'output_inds' must be provided if 'ts_inds' has hyper-indices.")
        output_inds = [x for x, c in left.items() if c == 0]
    output = frozenset(output_inds)
    if not output <= {x for xs in ts for x in xs}:
        raise ValueError("'output_inds' is not consistent with 'ts_inds'.")
    for a, b in path:
        a, b = sorted((a, b))
        if a == b:
            raise ValueError("'path' is not valid.")
        yb = ts.pop(b)
        xa = ts.pop(a)
        shared = frozenset(xa) & frozenset(yb)
        stay = frozenset(x for x in shared if left[x] > 1) | (output & shared)
        for x in shared:
            left[x] -= 1
        ts.append(tuple(x for x in xa if x in stay) + tuple(x for x in xa if x not in shared)
                  + tuple(y for y in yb if y not in shared))
    return ts, output & {x for xs in ts for x in xs}


_LINE = re.compile(r"^\d+(\s+\S+)*\s*$")


def load_tn(obj: Any, *, fuse=4, decompose_hyper_inds: bool = True, output_index_token="*",
            sparse_index_token="/", seed=None, **unsupported) -> TensorNetwork:
    """Index-list loader (the `load_tn` cases of tnco/app/app.py:438-492) + pre-fusing.

    `fuse` (default 4, app.py:156): tensors are pre-contracted at random while no intermediate
    exceeds that width; `tn.tags['fuse_path']` holds the path (app.py:373-414).  Arrays are not
    part of this build, so hyper-indices are never decomposed: like the reference without arrays
    (app.py:339-343) a warning is emitted when the network has hyper-indices.
    """
    unsupported = {k: v for k, v in unsupported.items()
                   if k not in ("atol", "dtype", "backend", "verbose", "simplify_circuit",
                                "initial_state", "final_state")}
    if unsupported:
        raise TypeError(f"Got unexpected keyword arguments: {sorted(unsupported)}")
    if isinstance(obj, TensorNetwork):
        return _fused(obj, fuse, decompose_hyper_inds, seed)
    if isinstance(obj, str):
        lines = [ln for ln in obj.splitlines() if ln.strip() and not ln.lstrip().startswith("#")]
        if not lines or not all(_LINE.match(ln.strip()) for ln in lines):
            raise TypeError("'obj' is not recognized.")
        obj = [(int(d), *ts) for d, *ts in (re.sub(r"\s+", " ", ln).strip().split() for ln in lines)]
    try:
        ok = all(len(x) > 1 and int(x[0]) == x[0] for x in obj)
    except (TypeError, ValueError):
        ok = False
    if not ok:
        raise TypeError("'obj' is not recognized.")
    tensor_map, dims, output_inds, sparse_inds = read_inds(
        dict(enumerate(obj)), output_index_token=output_index_token, sparse_index_token=sparse_index_token)
    tn = TensorNetwork((Tensor(xs, [dims[x] for x in xs], tags=dict(name=name)) for name, xs in tensor_map.items()),
                       output_inds=output_inds, sparse_inds=sparse_inds)
    return _fused(tn, fuse, decompose_hyper_inds, seed)


def _fused(tn: TensorNetwork, fuse_width, decompose_hyper_inds, seed) -> TensorNetwork:
    """The TensorNetwork branch of load_tn (tnco/app/app.py:314-420) for networks without arrays."""
    import warnings
    if tn.sparse_inds and (decompose_hyper_inds or fuse_width):
        warnings.warn("The decomposition of hyper-indices and the fusion of indices is not yet supported "
                      "if there are sparse indices")
        decompose_hyper_inds, fuse_width = False, False
    if decompose_hyper_inds and len(tn):
        warnings.warn("Cannot decompose hyper-indices if not all arrays are provided.")
    if fuse_width is None or not fuse_width > 0:
        return tn
    if "fuse_path" in tn.tags:
        raise ValueError("'TensorNetwork' has already the tag 'fuse_path'.")
    dims = tn.dims
    path = fuse(tn.ts_inds, dims, max_width=fuse_width, output_inds=tn.output_inds, seed=seed)
    ts_inds, output_inds = contract(path, tn.ts_inds, tn.output_inds, dims=dims)
    ts_tags = [t.tags or None for t in tn.tensors]
    for a, b in map(sorted, path):
        tb = ts_tags.pop(b)
        ta = ts_tags.pop(a)
        ts_tags.append(tb if ta is None else ta if tb is None else dict(x=ta, y=tb))
    tags = dict(tn.tags)
    tags["fuse_path"] = path
    return TensorNetwork((Tensor(xs, [dims[x] for x in xs], tags=tg or {}) for xs, tg in zip(ts_inds, ts_tags)),
                         output_inds=output_inds, sparse_inds=tn.sparse_inds, tags=tags)
