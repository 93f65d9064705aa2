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

__all__ = ["random_regular_tn", "random_hyper_tn", "chain_tn"]


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
