"""Minimal tensor-network container + loader for the optimizer front door.

Host mirror of the slice of /root/reference/tnco/app/tn.py:76-362 (Tensor,
TensorNetwork) and tnco/app/app.py:154-500 (load_tn) that feeds the SA path:
index-list inputs (list or string form, `*` output token, `/` sparse token;
tnco/utils/tn.py:520-569 read_inds).  Circuits, arrays, hyper-index
decomposition and pre-fusing (`fuse`) belong to the reference's front-end and
are out of scope: asking for them raises
NotImplementedError instead of silently doing something else.
"""
from __future__ import annotations

import json
import re
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Iterable

__all__ = ["Tensor", "TensorNetwork", "load_tn", "read_inds", "get_connected_components"]


@dataclass(frozen=True)
class Tensor:
    inds: tuple
    dims: tuple
    tags: dict = field(default_factory=dict)

    def __post_init__(self):
        object.__setattr__(self, "inds", tuple(self.inds))
        object.__setattr__(self, "dims", tuple(int(d) for d in self.dims))
        if len(self.inds) != len(self.dims):
            raise ValueError("Wrong number of 'inds'.")
        if any(d < 1 for d in self.dims):
            raise ValueError("Every dimension must be a positive integers.")

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
    """index -> (dim, tensor names...)  ==>  tensor map, dims, output, sparse (tn.py:520-569)."""
    if output_index_token == sparse_index_token:
        raise ValueError("'output_index_token' and 'sparse_index_token' must differ.")
    tensor_map = defaultdict(list)
    dims = {}
    for i, (d, *ts) in inds_map.items():
        dims[i] = int(d)
        for t in ts:
            tensor_map[t].append(i)
    output_inds = frozenset(tensor_map.pop(output_index_token, ()))
    sparse_inds = frozenset(tensor_map.pop(sparse_index_token, ()))
    return {k: tuple(v) for k, v in tensor_map.items()}, dims, output_inds, sparse_inds


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


_LINE = re.compile(r"^\d+(\s+\S+)*\s*$")


def load_tn(obj: Any, *, fuse=None, decompose_hyper_inds: bool = False, output_index_token="*",
            sparse_index_token="/", **unsupported) -> TensorNetwork:
    """Index-list loader (the `load_tn` cases of tnco/app/app.py:438-492).

    `fuse` (default 4 in the reference, app.py:156) and `decompose_hyper_inds` are front-end
    transformations outside this build's scope: only falsy values are accepted.
    """
    unsupported = {k: v for k, v in unsupported.items()
                   if k not in ("atol", "dtype", "backend", "seed", "verbose", "simplify_circuit",
                                "initial_state", "final_state")}
    if unsupported:
        raise TypeError(f"Got unexpected keyword arguments: {sorted(unsupported)}")
    if fuse:
        raise NotImplementedError("pre-fusing tensors (fuse > 0) is not part of this build; pass fuse=None.")
    if decompose_hyper_inds:
        raise NotImplementedError("hyper-index decomposition is not part of this build.")
    if isinstance(obj, TensorNetwork):
        return obj
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
    return TensorNetwork((Tensor(xs, [dims[x] for x in xs], tags=dict(name=name)) for name, xs in tensor_map.items()),
                         output_inds=output_inds, sparse_inds=sparse_inds)
