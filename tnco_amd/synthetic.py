"""Synthetic tensor networks for the benchmark and parity tests.

BASELINE.json's configs are quoted on "random-regular" tensor networks: tensors
= vertices of a random 3-regular simple connected graph, indices = edges, all
dimensions 2, no output indices.  The generator below
is the build's own (pairing model with rejection; deterministic for a given
seed through numpy's frozen RandomState stream) -- the reference has no
equivalent besides tnco/testing/utils.py:183-359 (random hypergraphs).
"""
from __future__ import annotations

import numpy as np

__all__ = ["random_regular_tn", "random_hyper_tn", "chain_tn", "sycamore53_tn", "sycamore53_cz_tn", "Problem",
           "regular_problem", "sycamore_problem", "replica_seeds", "linear_betas"]


def _connected(n: int, edges: np.ndarray) -> bool:
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    comps = n
    for a, b in edges:
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[ra] = rb
            comps -= 1
    return comps == 1


def random_regular_tn(n: int, degree: int = 3, seed: int = 0):
    """Random `degree`-regular simple connected graph as a tensor network.

    Returns:
        (ts_inds, dims, output_inds): ts_inds[v] is the sorted list of edge ids
        incident to vertex v (edge ids follow the lexicographic order of
        (min(u,v), max(u,v))); dims = 2; output_inds = ().
    """
    if (n * degree) % 2 or degree >= n:
        raise ValueError("n * degree must be even and degree < n.")
    rng = np.random.RandomState(seed)
    while True:
        stubs = np.repeat(np.arange(n), degree)
        rng.shuffle(stubs)
        e = np.sort(stubs.reshape(-1, 2), axis=1)
        if np.any(e[:, 0] == e[:, 1]):
            continue
        key = e[:, 0].astype(np.int64) * n + e[:, 1]
        if len(np.unique(key)) != len(key):
            continue
        e = e[np.argsort(key)]
        if not _connected(n, e):
            continue
        break
    ts_inds = [[] for _ in range(n)]
    for k, (a, b) in enumerate(e):
        ts_inds[int(a)].append(k)
        ts_inds[int(b)].append(k)
    return ts_inds, 2, ()


def chain_tn(n: int):
    """Open chain of n tensors: index k joins tensors k and k+1 (README.md:93-106)."""
    ts_inds = [[] for _ in range(n)]
    for k in range(n - 1):
        ts_inds[k].append(k)
        ts_inds[k + 1].append(k)
    return ts_inds, 2, ()


# Coupler patterns of the 54-qubit lattice (9 rows x 6; qubit (r, j) at column 2 j + r % 2; a coupler joins
# (r, c) to (r + 1, c +- 1)).  Key: (row parity of the upper qubit, direction of the step down).
_SYCAMORE_LAYOUTS = {
    # the supremacy experiment (Arute et al., Nature 574, 505 (2019), Fig. 3 and the published circuit files
    # circuit_n53_m20_s*_e0_pABCDCDAB): in the rotated (cirq GridQubit) picture A, B are the two staggered
    # halves of ONE coupler orientation and C, D the halves of the other.  With (R, C) the grid coordinates,
    # r = R + C - 5 and c = C - R + 5 here: a "vertical" pair (R, C)-(R + 1, C) is a step down-LEFT, a
    # "horizontal" pair (R, C)-(R, C + 1) a step down-RIGHT; A = vertical with R + C even (r odd), B =
    # vertical with R + C odd (r even), C = horizontal with R + C odd (r even), D = horizontal with R + C even.
    "supremacy": {(1, -1): "A", (0, -1): "B", (0, +1): "C", (1, +1): "D"},
    # rounds 1-3 of this build: the two diagonal orientations ALTERNATE from cycle to cycle -- a much easier
    # network (greedy trees 5-7 orders of magnitude cheaper); kept for the regression fixtures made on it.
    "alternating": {(0, +1): "A", (0, -1): "B", (1, +1): "C", (1, -1): "D"},
}
# the qubit that did not work in the experiment: GridQubit(3, 2) -> (r, c) = (0, 4)
_SYCAMORE_DEAD = {"supremacy": (0, 4), "alternating": (0, 0)}


def sycamore53_tn(depth: int = 20, layout: str = "supremacy", dead=None, order: str = "ABCDCDAB"):
    """Tensor network of a Sycamore-53 random quantum circuit amplitude <x|U|0>.

    BASELINE config 5 names a "Sycamore-53 depth-20 RQC"; the reference ships no such network (its
    circuit front-end needs cirq), so the topology is generated here: 54 qubits on a rotated square
    lattice of 9 rows x 6 (qubit (r, j) at column 2j + r % 2, diagonal nearest neighbours, 88
    couplers), the dead qubit removed (53 qubits, 86 couplers); the couplers split into the four
    patterns A, B, C, D (`_SYCAMORE_LAYOUTS`); cycle k applies pattern order[k % 8] of the supremacy
    sequence A B C D C D A B; single-qubit gates are absorbed into the two-qubit gates.
    Tensors: one 1-leg tensor per qubit at the input and at the output, one 4-leg tensor per
    two-qubit gate; indices: the wire segments between consecutive tensors of a qubit, all of
    dimension 2.  layout "supremacy", depth 20 -> 430 two-qubit gates (the published count), 536
    tensors, 913 indices (15 mask words); layout "alternating" (the easier network of rounds 1-3):
    435 gates, 541 tensors, 923 indices.

    Returns (ts_inds, dims, output_inds) like random_regular_tn.
    """
    names = _SYCAMORE_LAYOUTS[layout]
    dead = _SYCAMORE_DEAD[layout] if dead is None else tuple(dead)
    qubits = [(r, 2 * j + (r % 2)) for r in range(9) for j in range(6) if (r, 2 * j + (r % 2)) != dead]
    qset = set(qubits)
    patterns = {"A": [], "B": [], "C": [], "D": []}
    for (r, c) in qubits:
        for dc in (+1, -1):
            other = (r + 1, c + dc)
            if other in qset:
                patterns[names[(r % 2, dc)]].append(((r, c), other))
    wire = {q: None for q in qubits}  # current open index of every qubit
    ts_inds, n_idx = [], 0

    def new_index():
        nonlocal n_idx
        n_idx += 1
        return n_idx - 1

    for q in qubits:  # |0> on every qubit
        wire[q] = new_index()
        ts_inds.append([wire[q]])
    for cycle in range(depth):
        for qa, qb in patterns[order[cycle % len(order)]]:
            ia, ib = wire[qa], wire[qb]
            oa, ob = new_index(), new_index()
            ts_inds.append([ia, ib, oa, ob])
            wire[qa], wire[qb] = oa, ob
    for q in qubits:  # <x| on every qubit
        ts_inds.append([wire[q]])
    return ts_inds, 2, ()


def sycamore53_cz_tn(depth: int = 12, fuse: float | None = 4, seed: int = 0, order: str = "ABCDCDAB"):
    """A hyper-index network as the reference's loader produces them: the Sycamore-53 lattice and coupler sequence
    with DIAGONAL two-qubit gates (CZ, as in the 2018 supremacy proposals) and a layer of non-diagonal single-qubit
    gates on every qubit before each cycle.  `decompose_hyper_inds` (tnco/utils/tn.py:827, on by default,
    tnco/app/app.py:157,351-358) turns a diagonal gate into a tensor whose input and output leg on a wire are ONE
    index: the CZ of qubits (a, b) is the 2-leg tensor [wire_a, wire_b], and a wire segment is an index held by the
    single-qubit gates at its ends and by every CZ in between -- 2 or 3 tensors here.  `fuse` (default 4, the
    loader's default, tnco/app/app.py:156): the random pre-contraction of tnco/utils/tn.py:598-824 (restated in
    tnco_amd/app/tn.py) applied with `seed`; None: the raw network.

    Returns (ts_inds, dims, output_inds); amplitude: no output indices."""
    names = _SYCAMORE_LAYOUTS["supremacy"]
    dead = _SYCAMORE_DEAD["supremacy"]
    qubits = [(r, 2 * j + (r % 2)) for r in range(9) for j in range(6) if (r, 2 * j + (r % 2)) != dead]
    qset = set(qubits)
    patterns = {"A": [], "B": [], "C": [], "D": []}
    for (r, c) in qubits:
        for dc in (+1, -1):
            other = (r + 1, c + dc)
            if other in qset:
                patterns[names[(r % 2, dc)]].append(((r, c), other))
    ts_inds, n_idx, wire = [], 0, {}
    for q in qubits:  # |0>
        wire[q] = n_idx
        ts_inds.append([n_idx])
        n_idx += 1
    for cycle in range(depth):
        for q in qubits:  # non-diagonal single-qubit gate: a new wire segment
            ts_inds.append([wire[q], n_idx])
            wire[q] = n_idx
            n_idx += 1
        for qa, qb in patterns[order[cycle % len(order)]]:  # CZ, diagonal: both wires pass through
            ts_inds.append([wire[qa], wire[qb]])
    for q in qubits:  # <x|
        ts_inds.append([wire[q]])
    if fuse is not None:
        from .app import tn as apptn
        path = apptn.fuse(ts_inds, 2, fuse, output_inds=(), seed=seed)
        ts_inds = [list(x) for x in apptn.contract(path, ts_inds, output_inds=(), dims=None)[0]] if path else ts_inds
        used = sorted({i for t in ts_inds for i in t})
        ren = {i: k for k, i in enumerate(used)}
        ts_inds = [[ren[i] for i in t] for t in ts_inds if t] + [t for t in ts_inds if not t]
    return ts_inds, 2, ()


def random_hyper_tn(n: int, n_inds: int, k: int = 3, n_output: int = 0, seed: int = 0,
                    dims_choices=(2,)):
    """Connected random hypergraph TN: every index sits on 2..k tensors.

    In the spirit of tnco/testing/utils.py:183-359 (k-uniform random hypergraph
    with guaranteed connectivity, optional output indices); own construction.
    Returns (ts_inds, dims(list per index), output_inds).
    """
    rng = np.random.RandomState(seed)
    ts_inds = [[] for _ in range(n)]
    # spanning structure first: index i joins tensor i+1 to a random earlier one
    idx = 0
    order = rng.permutation(n)
    for j in range(1, n):
        a = int(order[j])
        b = int(order[rng.randint(0, j)])
        members = {a, b}
        extra = rng.randint(0, k - 1)
        while len(members) < min(n, 2 + extra):
            members.add(int(rng.randint(0, n)))
        for t in sorted(members):
            ts_inds[t].append(idx)
        idx += 1
    while idx < n_inds:
        m = int(rng.randint(2, k + 1))
        members = set()
        while len(members) < min(n, m):
            members.add(int(rng.randint(0, n)))
        for t in sorted(members):
            ts_inds[t].append(idx)
        idx += 1
    n_inds = idx
    output = sorted(int(x) for x in rng.choice(n_inds, size=min(n_output, n_inds), replace=False)) if n_output else []
    dims = [int(dims_choices[int(rng.randint(0, len(dims_choices)))]) for _ in range(n_inds)]
    return ts_inds, dims, tuple(output)


class Problem:
    """A synthetic tensor network flattened to bit positions (what `tnco_hip_create` takes)."""

    def __init__(self, ts_inds, dims, output_inds=(), sparse_inds=(), n_inds=None):
        from . import ctree as ct
        self.ts_inds = [list(x) for x in ts_inds]
        self.n = len(self.ts_inds)
        self.n_inds = (max((max(x) for x in self.ts_inds if x), default=-1) + 1) if n_inds is None else n_inds
        self.W = ct.n_words(self.n_inds)
        self.dims = dims
        self.leaf_masks = ct.pack_masks(self.ts_inds, self.n_inds)
        self.output_mask = ct.pack_masks([list(output_inds)], self.n_inds)[0]
        self.sparse_mask = ct.pack_masks([list(sparse_inds)], self.n_inds)[0] if len(sparse_inds) else None
        self.holders = [[] for _ in range(self.n_inds)]
        for t, xs in enumerate(self.ts_inds):
            for i in xs:
                self.holders[i].append(t)

    def tree(self, seed):
        """Python spec of the native initial-tree generator (random Kruskal order)."""
        from . import ctree as ct
        con = ct.random_contraction(self.holders, self.n, seed=int(seed) & 0xFFFFFFFF)
        return ct.tree_from_contraction(con, self.n)

    def links(self, seeds):
        out = np.empty((len(seeds), 3, 2 * self.n - 1), np.int32)
        for k, s in enumerate(seeds):
            out[k, 0], out[k, 1], out[k, 2] = self.tree(s)
        return out

    def node_masks(self, left, right):
        from . import ctree as ct
        return ct.derive_inds(left, right, self.leaf_masks, self.output_mask)


def regular_problem(n, graph_seed, degree=3):
    ts, d, out = random_regular_tn(n, degree, graph_seed)
    return Problem(ts, d, out)


def sycamore_problem(depth=20, layout="supremacy"):
    ts, d, out = sycamore53_tn(depth, layout)
    return Problem(ts, d, out)


def replica_seeds(R, S=0):
    """seeds = Random(S).choices(range(2**32), k=R) -- tnco/app/infinite_memory/sa.py:237."""
    import random
    return random.Random(S).choices(range(2**32), k=R)


def linear_betas(b0, b1, n_steps):
    """more_itertools.numeric_range(b0, b1, (b1-b0)/n_steps): b0 + k*step (sa.py:155)."""
    step = (b1 - b0) / n_steps
    return np.array([b0 + k * step for k in range(n_steps)], np.float64)
