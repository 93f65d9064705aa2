"""Contraction trees flattened for the GPU: host-side construction and read-back.

Mirrors the reference's Python layer around the native tree
(/root/reference/tnco/ctree.py:38-388): a linear (einsum) contraction path is
turned into SSA triples, renumbered leaves-first, and stored as three int32
link arrays (left/right/parent, null = -1, leaves in [0, n), root last) plus
one leg bitmask per node packed in uint64 words (bit p of word p // 64 <->
index position p, the boost::dynamic_bitset convention of
include/tnco/bitset.hpp).  `path()` converts a node array back into the linear
format (ctree.py:350-388 over include/tnco/utils.hpp:34-71).
"""
from __future__ import annotations

import math
from collections import Counter
from typing import Any, Iterable, Sequence

import numpy as np

__all__ = [
    "ContractionTree", "pack_masks", "unpack_mask", "traverse", "get_contraction",
    "tree_from_contraction", "derive_inds", "random_contraction", "linear_to_ssa",
    "ssa_to_linear",
]

NULL = -1


def n_words(n_inds: int) -> int:
    return max(1, (int(n_inds) + 63) // 64)


def pack_masks(positions: Iterable[Iterable[int]], n_inds: int) -> np.ndarray:
    """List of index-position lists -> (len, W) uint64 masks."""
    positions = [list(p) for p in positions]
    W = n_words(n_inds)
    out = np.zeros((len(positions), W), np.uint64)
    for r, ps in enumerate(positions):
        for p in ps:
            if not 0 <= p < max(n_inds, 1):
                raise ValueError("index position out of range.")
            out[r, p >> 6] |= np.uint64(1) << np.uint64(p & 63)
    return out


def unpack_mask(mask: np.ndarray) -> list[int]:
    """One (W,) uint64 mask -> ascending list of set positions."""
    out = []
    for w, x in enumerate(np.asarray(mask, np.uint64).tolist()):
        while x:
            b = (x & -x).bit_length() - 1
            out.append(w * 64 + b)
            x &= x - 1
    return out


def traverse(left: Sequence[int], right: Sequence[int]) -> list[int]:
    """Post-order visit, child 0 subtree first (include/tnco/utils.hpp:34-51)."""
    N = len(left)
    stack = [N - 1]
    visited = [False] * N
    order = []
    while stack:
        pos = stack[-1]
        if visited[pos] or left[pos] < 0:
            stack.pop()
            order.append(pos)
        else:
            visited[pos] = True
            stack.append(int(right[pos]))
            stack.append(int(left[pos]))
    return order


def get_contraction(left: Sequence[int], right: Sequence[int]) -> list[tuple[int, int, int]]:
    """(child0, child1, node) per internal node in post-order (utils.hpp:53-71)."""
    return [(int(left[p]), int(right[p]), int(p)) for p in traverse(left, right) if left[p] >= 0]


def tree_from_contraction(contraction: Sequence[tuple[int, int, int]], n_leaves: int):
    """SSA triples over ids 0..2n-2 (leaves first, creation order) -> link arrays."""
    N = 2 * n_leaves - 1
    left = np.full(N, NULL, np.int32)
    right = np.full(N, NULL, np.int32)
    parent = np.full(N, NULL, np.int32)
    for x, y, z in contraction:
        left[z], right[z] = x, y
        parent[x] = z
        parent[y] = z
    return left, right, parent


def derive_inds(left, right, leaf_masks: np.ndarray, output_mask: np.ndarray | None = None,
                check_shared_inds: bool = False) -> np.ndarray:
    """Leg masks of every node from the leaves' masks.

    Same result as the hyper-count bookkeeping of ctree.py:163-189: the legs of
    z = (x, y) are x ^ y plus the shared legs still held by a tensor outside z
    or by the output.
    """
    N = len(left)
    n = (N + 1) // 2
    W = leaf_masks.shape[1]
    order = traverse(left, right)
    union = np.zeros((N, W), np.uint64)
    union[:n] = leaf_masks
    for p in order:
        if left[p] >= 0:
            union[p] = union[left[p]] | union[right[p]]
    outside = np.zeros((N, W), np.uint64)
    if output_mask is not None:
        outside[N - 1] = output_mask
    inds = np.zeros((N, W), np.uint64)
    inds[:n] = leaf_masks
    for p in reversed(order):
        if left[p] >= 0:
            a, b = left[p], right[p]
            outside[a] = outside[p] | union[b]
            outside[b] = outside[p] | union[a]
    for p in order:
        if left[p] >= 0:
            a, b = inds[left[p]], inds[right[p]]
            if check_shared_inds and not np.any(a & b):
                raise ValueError("'check_shared_inds' failed.")
            inds[p] = (a ^ b) | (a & b & outside[p])
    return inds


def linear_to_ssa(path: Iterable[tuple[int, int]], n_tensors: int) -> list[tuple[int, int, int]]:
    """Linear einsum path -> SSA triples (ctree.py:113-122)."""
    pos = list(range(n_tensors))
    out = []
    for i, xs in enumerate(path):
        x, y = sorted(xs)
        py = pos.pop(y)
        px = pos.pop(x)
        pos.append(i + n_tensors)
        out.append((px, py, pos[-1]))
    return out


def ssa_to_linear(contraction: Iterable[tuple[int, int, int]], n_tensors: int) -> list[tuple[int, int]]:
    """SSA triples -> linear einsum path (ctree.py:371-388)."""
    all_pos = list(range(n_tensors))
    path = []
    for x, y, z in contraction:
        p = (all_pos.index(x), all_pos.index(y))
        path.append(p)
        if p[0] > p[1]:
            p = (p[1], p[0])
        all_pos.pop(p[1])
        all_pos.pop(p[0])
        all_pos.append(z)
    return path


def _mt_raw(seed: int, k: int) -> np.ndarray:
    """First k raw outputs of std::mt19937 seeded with `seed` (numpy's legacy
    RandomState seeds MT19937 with the same init_genrand recurrence)."""
    rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    return rs._bit_generator.random_raw(k).astype(np.uint64)


def random_contraction(holders: Sequence[Sequence[int]], n_tensors: int, seed: int):
    """Seeded random initial contraction of one CONNECTED component.

    The build's own initial-tree generator (the reference calls opt_einsum's
    greedy on a shuffled tensor list, tnco/utils/tn.py:195-230; opt_einsum is
    an unpinned third-party dependency, so the initial tree is outside the
    parity contract and both the oracle and the GPU path are fed this one).

    Random-Kruskal: indices are visited in a seeded random order (Fisher-Yates
    driven by mt19937(seed): for i = I-1..1 swap(perm[i], perm[raw_k % (i+1)]));
    each index merges, in ascending tensor order, the components of the
    tensors that hold it.  Every merge contracts two tensors sharing that
    index, so `check_shared_inds` holds (tnco/app/infinite_memory/sa.py:186-190).
    The native batched twin is tnco_hip_random_trees (csrc/host_trees.cpp).

    Args:
        holders: holders[i] = ascending tensor ids (local to the component,
            0..n_tensors-1) holding index i.
    Returns:
        SSA triples (child0, child1, new) with child0 < child1.
    """
    I = len(holders)
    raw = _mt_raw(seed, max(I - 1, 0))
    perm = list(range(I))
    k = 0
    for i in range(I - 1, 0, -1):
        j = int(raw[k] % np.uint64(i + 1))
        k += 1
        perm[i], perm[j] = perm[j], perm[i]
    uf = list(range(n_tensors))
    node = list(range(n_tensors))

    def find(a):
        while uf[a] != a:
            uf[a] = uf[uf[a]]
            a = uf[a]
        return a

    out = []
    nxt = n_tensors
    for idx in perm:
        hs = holders[idx]
        if len(hs) < 2:
            continue
        for t in hs[1:]:
            ra, rb = find(hs[0]), find(t)
            if ra == rb:
                continue
            a, b = node[ra], node[rb]
            out.append((min(a, b), max(a, b), nxt))
            r = min(ra, rb)
            uf[ra] = r
            uf[rb] = r
            node[r] = nxt
            nxt += 1
    if len(out) != n_tensors - 1:
        raise ValueError("tensor network component is not connected.")
    return out


def ssa_greedy(inputs: Sequence[frozenset], output: frozenset) -> list[tuple[int, int]]:
    """opt_einsum's greedy path finder on index sets of dimension 2, SSA form.

    The reference gets its initial path from `oe.contract_path(subscripts, *shapes, shapes=True,
    optimize='greedy')` with every shape (2,) * rank (tnco/utils/tn.py:197-216).  opt_einsum is a
    third-party dependency that is NOT pinned by the reference (pyproject.toml:55) and is absent
    from this image, so this is a restatement of its published algorithm (opt_einsum 3.3/3.4
    `paths.ssa_greedy_optimize` with `choose_fn=_simple_chooser`, `cost_fn='memory-removed'`):
    **parity unpinned** -- it cannot be checked against opt_einsum here.

    1. dims common to all inputs join the output; inputs with identical index sets are multiplied
       eagerly (keys of `remaining` are the index SETS);
    2. candidates (cost, k1, k2, k12) with cost = (size(k12) - size(k1) - size(k2), id2, id1),
       id1 < id2 the ssa ids at push time; k12 = (either & output) | (two & dims held by >= 3
       keys) | (one & dims held by >= 2 keys); per new tensor only the cheapest candidate among its
       neighbours is pushed; a popped candidate is obsolete when one of its keys is gone;
    3. what is left is combined by outer products, smallest output size first.
    The native batched twin is csrc/host_trees.cpp (mode TNCO_HIP_TREES_GREEDY), tested equal.
    """
    import heapq
    import itertools
    if len(inputs) == 1:
        return []
    fs_inputs = [frozenset(x) for x in inputs]
    output = frozenset(output) | frozenset.intersection(*fs_inputs)
    size = lambda key: 1 << len(key)  # noqa: E731  (every dimension is 2)

    remaining: dict[frozenset, int] = {}
    ssa_ids = itertools.count(len(fs_inputs))
    ssa_path = []
    for ssa_id, key in enumerate(fs_inputs):
        if key in remaining:
            ssa_path.append((remaining[key], ssa_id))
            remaining[key] = next(ssa_ids)
        else:
            remaining[key] = ssa_id

    dim_to_keys: dict[Any, set] = {}
    for key in remaining:
        for dim in key - output:
            dim_to_keys.setdefault(dim, set()).add(key)
    ref = {c: set(d for d, keys in dim_to_keys.items() if len(keys) >= c) - output for c in (2, 3)}
    footprints = {key: size(key) for key in remaining}

    def candidate(k1, k2):
        either, two = k1 | k2, k1 & k2
        one = either - two
        k12 = (either & output) | (two & ref[3]) | (one & ref[2])
        cost = size(k12) - footprints[k1] - footprints[k2]
        id1, id2 = remaining[k1], remaining[k2]
        if id1 > id2:
            k1, id1, k2, id2 = k2, id2, k1, id1
        return (cost, id2, id1), k1, k2, k12

    def push(k1, k2s, queue):
        # (min over the cost triples: they are distinct, so no index set is ever compared)
        heapq.heappush(queue, _Cand(min((candidate(k1, k2) for k2 in k2s), key=lambda c: c[0])))

    queue: list = []
    for dim, dim_keys in dim_to_keys.items():
        lst = sorted(dim_keys, key=remaining.__getitem__)
        for i, k1 in enumerate(lst[:-1]):
            push(k1, lst[1 + i:], queue)

    while queue:
        _cost, k1, k2, k12 = heapq.heappop(queue).c
        if k1 not in remaining or k2 not in remaining:
            continue
        id1, id2 = remaining.pop(k1), remaining.pop(k2)
        for dim in k1 - output:
            dim_to_keys[dim].remove(k1)
        for dim in k2 - output:
            dim_to_keys[dim].remove(k2)
        ssa_path.append((id1, id2))
        if k12 in remaining:
            ssa_path.append((remaining[k12], next(ssa_ids)))
        else:
            for dim in k12 - output:
                dim_to_keys.setdefault(dim, set()).add(k12)
        remaining[k12] = next(ssa_ids)
        for dim in k1 | k2 - output:  # (sic: k1 | (k2 - output), as published)
            count = len(dim_to_keys.get(dim, ()))
            if count <= 1:
                ref[2].discard(dim)
                ref[3].discard(dim)
            elif count == 2:
                ref[2].add(dim)
                ref[3].discard(dim)
            else:
                ref[2].add(dim)
                ref[3].add(dim)
        footprints[k12] = size(k12)
        k2s = set(k2 for dim in k12 - output for k2 in dim_to_keys[dim])
        k2s.discard(k12)
        if k2s:
            push(k12, k2s, queue)

    heap = [(size(key & output), ssa_id, key) for key, ssa_id in remaining.items()]
    heap = [_Cand(x) for x in heap]
    heapq.heapify(heap)
    _, id1, k1 = heapq.heappop(heap).c
    while heap:
        _, id2, k2 = heapq.heappop(heap).c
        ssa_path.append((min(id1, id2), max(id1, id2)))
        k12 = (k1 | k2) & output
        _, id1, k1 = heapq.heappushpop(heap, _Cand((size(k12), next(ssa_ids), k12))).c
    return ssa_path


class _Cand:
    """Heap entry ordered by its leading fields only: opt_einsum compares whole tuples, whose
    leading (cost, id2, id1) / (size, ssa id) fields are distinct for distinct contractions; equal
    ones are the same contraction pushed twice, and then the order does not matter."""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = c

    def __lt__(self, other):
        a, b = self.c, other.c
        return (a[0], a[1]) < (b[0], b[1]) if not isinstance(a[0], tuple) else a[0] < b[0]


def greedy_contraction(leaf_positions: Sequence[Sequence[int]], output_positions: Iterable[int], seed: int,
                       rng=None):
    """Initial contraction of one connected component as the reference draws it
    (tnco/utils/tn.py:189-230): `Random(seed).shuffle` of the component's tensors, opt_einsum's
    greedy on the shuffled list with every dimension 2, back to tensor ids.

    Args:
        leaf_positions: index positions of the component's tensors, in ascending tensor order.
        output_positions: output indices that survive the reference's filter (held by at most one
            tensor, tn.py:175-178); the others are contractible edges.
        seed: the run's seed (also the optimizer's, sa.py:176,193).
        rng: a `random.Random` already advanced by the components before this one (the reference
            shares ONE generator over the components, tn.py:163,192); default `Random(seed)`.
    Returns:
        SSA triples (child0, child1, new) with child0 < child1, leaves 0..n-1.
    """
    from random import Random
    n = len(leaf_positions)
    if rng is None:
        rng = Random(seed)
    order = list(range(n))
    rng.shuffle(order)
    inputs = [frozenset(leaf_positions[t]) for t in order]
    out = frozenset(output_positions) & frozenset().union(*inputs) if inputs else frozenset()
    if n == 2 or (n > 2 and frozenset().union(*inputs) == out):
        ssa = [(0, 1)] if n == 2 else None  # contract_path short-cuts (<= 2 operands / nothing to sum)
        if ssa is None:
            raise NotImplementedError("all indices are output indices: a single n-ary contraction in opt_einsum.")
    else:
        ssa = ssa_greedy(inputs, out)
    node = lambda x: order[x] if x < n else x  # noqa: E731
    return [(min(node(a), node(b)), max(node(a), node(b)), n + s) for s, (a, b) in enumerate(ssa)]


class ContractionTree:
    """Flattened contraction tree (host mirror of tnco.ctree.ContractionTree).

    Args follow /root/reference/tnco/ctree.py:69-79.  Attributes: `left`,
    `right`, `parent` (int32[N]), `masks` (uint64[N, W]), `dims_vec`
    (per-position dims), `inds_order` (position -> index name), `tensors_pos`.
    """

    def __init__(self, path: Iterable[tuple[int, int]], ts_inds: Iterable[Iterable[Any]],
                 dims: dict[Any, int] | int, *, output_inds: Iterable[Any] | None = None,
                 check_shared_inds: bool = False):
        ts_inds = [list(xs) for xs in ts_inds]
        n_tensors = len(ts_inds)
        contraction = linear_to_ssa(path, n_tensors)
        flat = [p for xs in contraction for p in xs]
        self._n_tensors = n_tensors
        self._tensors_pos = tuple(sorted({p for p in flat if p < n_tensors}))
        all_inds = list(dict.fromkeys(i for t in self._tensors_pos for i in ts_inds[t]))
        hyper_count = {x: c - 1 for x, c in Counter(
            i for t in self._tensors_pos for i in ts_inds[t]).items()}
        if output_inds is None:
            if any(c > 1 for c in hyper_count.values()):
                raise ValueError("'output_inds' must be provided if 'ts_inds' "
                                 "has hyper-indices.")
            output = frozenset(x for x, c in hyper_count.items() if c == 0)
        else:
            output = frozenset(output_inds)
        output = output.intersection(all_inds)
        for x in output:
            hyper_count[x] += 1
        if not contraction:
            raise ValueError("'path' cannot be empty.")
        ext = ts_inds + [None] * (max(flat) - n_tensors + 1)
        for tx, ty, tz in contraction:
            ix, iy = frozenset(ext[tx]), frozenset(ext[ty])
            shared = ix & iy
            if check_shared_inds and not shared:
                raise ValueError("'check_shared_inds' failed.")
            iz = set(ix ^ iy)
            for s in shared:
                hyper_count[s] -= 1
                if hyper_count[s] > 0:
                    iz.add(s)
            # keep a deterministic leg order (the reference's tuple(set) order is
            # hash-dependent; only membership reaches the masks)
            ext[tz] = tuple(x for x in list(dict.fromkeys(list(ext[tx]) + list(ext[ty]))) if x in iz)
        pos = sorted(set(flat))
        tree_map = {p: k for k, p in enumerate(pos)}
        tree = [tuple(tree_map[p] for p in xs) for xs in contraction]
        n_leaves = len(self._tensors_pos)
        self.left, self.right, self.parent = tree_from_contraction(tree, n_leaves)
        node_inds = [ext[p] for p in pos]
        self.inds_order = tuple(dict.fromkeys(i for xs in node_inds for i in xs))
        imap = {x: k for k, x in enumerate(self.inds_order)}
        self.n_inds = len(self.inds_order)
        self.masks = pack_masks([[imap[i] for i in xs] for xs in node_inds], self.n_inds)
        try:
            d = int(dims)
            if d != dims:
                raise ValueError("'dims' is not valid.")
            self.dims_vec = np.full(self.n_inds, d, np.uint64)
        except TypeError:
            self.dims_vec = np.array([int(dims[x]) for x in self.inds_order], np.uint64)
        self.output_mask = pack_masks([[imap[i] for i in output if i in imap]], self.n_inds)[0]

    # -- mirrors of the reference accessors ---------------------------------
    def __len__(self) -> int:
        return len(self.left)

    @property
    def n_leaves(self) -> int:
        return (len(self.left) + 1) // 2

    @property
    def tensors_pos(self) -> tuple[int, ...]:
        return self._tensors_pos

    @property
    def dims(self) -> dict[Any, int]:
        return dict(zip(self.inds_order, (int(d) for d in self.dims_vec)))

    @property
    def inds(self) -> list[frozenset]:
        return [frozenset(self.inds_order[p] for p in unpack_mask(m)) for m in self.masks]

    def with_links(self, left, right, parent, masks=None) -> "ContractionTree":
        """Same leaves / index naming, different tree (e.g. a replica's best tree)."""
        other = object.__new__(ContractionTree)
        other.__dict__.update(self.__dict__)
        other.left = np.asarray(left, np.int32).copy()
        other.right = np.asarray(right, np.int32).copy()
        other.parent = np.asarray(parent, np.int32).copy()
        other.masks = (derive_inds(other.left, other.right, self.masks[:self.n_leaves], self.output_mask)
                       if masks is None else np.asarray(masks, np.uint64).copy())
        return other

    def path(self) -> list[tuple[int, int]]:
        """Linear (einsum) path over the ORIGINAL tensor list (ctree.py:350-388)."""
        shift = self._n_tensors - self.n_leaves

        def rescale(p):
            return self._tensors_pos[p] if p < len(self._tensors_pos) else p + shift

        contraction = [tuple(rescale(p) for p in xs) for xs in get_contraction(self.left, self.right)]
        return ssa_to_linear(contraction, self._n_tensors)

    def max_width(self) -> float:
        return max(math.log2(math.prod(int(self.dims_vec[p]) for p in unpack_mask(m))) if np.any(m) else 0.0
                   for m in self.masks)
