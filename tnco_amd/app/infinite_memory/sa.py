"""method='sa', infinite memory: the SA driver behind the plugin API.

Host mirror of /root/reference/tnco/app/infinite_memory/sa.py:63-257.  What the reference does
per run in a loky worker process (random initial path -> ContractionTree -> Optimizer ->
`for beta in betas: opt.update(prob)`) is done here for ALL runs at once on the GPU: one
BatchedOptimizer per connected component, one kernel launch per chunk of the beta schedule.
With torch.distributed initialised, the run list is sharded over the ranks (one GPU each,
tnco_amd/parallel.py) and merged with one all-reduce(min) + one all-gather of the heads.

Differences a caller can observe:
  * the initial tree of a run comes from this build's generator, not opt_einsum's greedy
    (third-party, unpinned in the reference), so costs differ from the reference run by run
    while following the same distribution of moves;
  * `load_tn` takes index lists only and does not pre-fuse (`fuse` defaults to None, the reference
    to 4);
  * only the `top_k` best runs (default min(n_runs, 1024)) are materialised as results.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from time import perf_counter
from typing import Any, Iterable

import numpy as np

from ... import core, parallel
from ...ctree import get_contraction, n_words, pack_masks, ssa_to_linear
from ..app import BaseContractionResults, BaseOptimizer, JSONEncoder as BaseJSONEncoder, cost_to_decimal
from ..tn import get_connected_components

__all__ = ["Optimizer", "ContractionResults", "merge_contraction_paths", "expand_betas"]


class JSONEncoder(BaseJSONEncoder):
    def default(self, obj):
        if isinstance(obj, ContractionResults):
            return dict(**BaseJSONEncoder().default(obj), disconnected_paths=obj.disconnected_paths)
        return super().default(obj)


@dataclass(repr=False, frozen=True, eq=False)
class ContractionResults(BaseContractionResults):
    """sa.py:63-90: + per-component costs and paths (each path over ALL original tensors)."""
    disconnected_costs: list
    disconnected_paths: list

    def to_json(self):
        return json.dumps(self, cls=JSONEncoder)


def merge_contraction_paths(n_tensors: int, paths: Iterable[list], *, autocomplete: bool = True) -> list:
    """tnco/utils/tn.py:334-401."""
    merged_pos = list(range(n_tensors))
    merged_path = []
    for i, path in enumerate(paths):
        pos = list(range(n_tensors))
        for x, y in path:
            x, y = sorted((x, y))
            y = pos.pop(y)
            x = pos.pop(x)
            pos.append((i, len(pos)))
            try:
                mx, my = sorted((merged_pos.index(x), merged_pos.index(y)))
            except ValueError as e:
                raise ValueError("'paths' are not valid or not disconnected.") from e
            merged_path.append((mx, my))
            merged_pos.pop(my)
            merged_pos.pop(mx)
            merged_pos.append(pos[-1])
    if autocomplete:
        merged_path += [(0, 1)] * (len(merged_pos) - 1)
    return merged_path


def expand_betas(betas, n_steps):
    """Argument checks and schedule of sa.py:141-156 (more_itertools.numeric_range(b0, b1, step)
    yields b0 + k*step while < b1 for step > 0, > b1 for step < 0)."""
    if n_steps is not None:
        if int(n_steps) != n_steps or n_steps <= 0:
            raise ValueError("'n_steps' must be a positive number.")
        n_steps = int(n_steps)
    if isinstance(betas, tuple) and len(betas) == 2:
        if n_steps is None:
            raise ValueError("'n_steps' must be provided if 'betas' has the format '(beta_min, beta_max)'.")
        if betas[0] == betas[1]:
            raise ValueError("'betas' must use the format '(beta_ini, beta_end)', with 'beta_ini != beta_end'.")
        b0, b1 = betas
        step = (b1 - b0) / n_steps
        out, k = [], 0
        while True:
            v = b0 + k * step
            if (step > 0 and v >= b1) or (step < 0 and v <= b1):
                break
            out.append(v)
            k += 1
        betas = out
    betas = [float(b) for b in betas]
    if n_steps is not None:
        betas = betas[:n_steps]  # `if n == n_steps: break`, sa.py:201
    return np.asarray(betas, np.float64)


class _Component:
    """One connected component flattened to bit positions."""

    def __init__(self, tn, cc):
        self.tensors = tuple(cc)
        ts = [tn.ts_inds[t] for t in cc]
        self.inds_order = tuple(dict.fromkeys(i for xs in ts for i in xs))
        imap = {x: k for k, x in enumerate(self.inds_order)}
        self.n_inds = len(self.inds_order)
        self.leaf_positions = [[imap[i] for i in xs] for xs in ts]
        self.leaf_masks = pack_masks(self.leaf_positions, self.n_inds)
        self.output_mask = pack_masks([[imap[i] for i in tn.output_inds if i in imap]], self.n_inds)[0]
        sp = [imap[i] for i in tn.sparse_inds if i in imap]
        self.sparse_mask = pack_masks([sp], self.n_inds)[0] if sp else None
        dims = [tn.dims[x] for x in self.inds_order]
        self.dims = dims[0] if dims and all(d == dims[0] for d in dims) else np.asarray(dims, np.uint64)
        if not dims:
            self.dims = 1

    def path(self, left, right, n_tensors):
        """Linear path over all original tensors (tnco/ctree.py:350-388)."""
        nc = len(self.tensors)
        shift = n_tensors - nc

        def rescale(p):
            return self.tensors[p] if p < nc else p + shift

        return ssa_to_linear([tuple(rescale(p) for p in xs) for xs in get_contraction(left, right)], n_tensors)


class Optimizer(BaseOptimizer):
    """Simulated annealing with no memory constraint, all runs on the GPU."""

    def optimize(self, tn: Any, betas, n_steps: int | None = None, n_runs: int = 1,
                 n_projs: int | None = None, timeout: float | None = None, *, top_k: int | None = None,
                 sweeps_per_launch: int = 100, prob: str = "mh", device: int | None = None,
                 **load_tn_options) -> Any:
        tn = self._load_tn(tn, **load_tn_options)
        betas = expand_betas(betas, n_steps)
        n_runs = int(n_runs)
        if n_runs <= 0:
            raise ValueError("'n_runs' must be a positive number.")
        if tn.sparse_inds and not n_projs:
            raise ValueError("'n_projs' must be provided if 'tn' has sparse indices.")
        seeds = self._rng.choices(range(2**32), k=n_runs)  # sa.py:237

        rank, world = parallel.rank_world()
        lo, hi = parallel.shard_bounds(n_runs, world, rank)
        if device is None:
            device = parallel.local_device()
        my_seeds = seeds[lo:hi]
        top_k = min(n_runs, 1024) if top_k is None else max(1, min(int(top_k), n_runs))

        comps = [_Component(tn, cc) for cc in get_connected_components(tn.ts_inds)]
        n_local = len(my_seeds)
        raw_cost = np.zeros((n_local, len(comps)), np.float64)
        handles = []
        t0 = perf_counter()
        timed_out = False
        for ci, comp in enumerate(comps):
            if len(comp.tensors) <= 1 or n_local == 0:  # `if not path`, sa.py:179-183
                handles.append(None)
                continue
            links = core.random_trees(comp.leaf_positions, comp.n_inds, my_seeds)
            opt = core.BatchedOptimizer(comp.leaf_masks, links, my_seeds, n_inds=comp.n_inds, dims=comp.dims,
                                        output_mask=comp.output_mask, sparse_mask=comp.sparse_mask,
                                        n_projs=n_projs, cost_type=self.cost_type, device=device)
            for s in range(0, len(betas), max(1, int(sweeps_per_launch))):
                if timeout is not None and perf_counter() - t0 > timeout:
                    timed_out = True
                    break
                opt.run(betas[s:s + sweeps_per_launch], prob)
                if timeout is not None:
                    opt.sync()
            raw_cost[:, ci] = opt.costs()[1]
            handles.append(opt)
        runtime = perf_counter() - t0

        # cost of a run = sum of the per-component Decimals (sa.py:215-218)
        dec = [[cost_to_decimal(raw_cost[r, ci]) if handles[ci] is not None else 0 for ci in range(len(comps))]
               for r in range(n_local)]
        totals = [sum(d) for d in dec]
        order = sorted(range(n_local), key=lambda r: (totals[r], lo + r))[:top_k]
        local = []
        for r in order:
            paths = []
            for ci, comp in enumerate(comps):
                if handles[ci] is None:
                    paths.append([])
                else:
                    l, rr, _p, _m = handles[ci].tree(r, which_min=True, with_masks=False)
                    paths.append(comp.path(l, rr, len(tn)))
            local.append((totals[r], lo + r, dec[r], paths))
        best_raw = float(raw_cost.sum(axis=1).min()) if n_local else float("inf")
        for h in handles:
            if h is not None:
                h.close()

        merged = parallel.merge_heads(local, top_k, rank, world)
        tn.tags["best_raw_cost"] = parallel.global_best(best_raw, rank, world, device)
        tn.tags["n_runs"] = n_runs
        tn.tags["timed_out"] = timed_out
        results = [ContractionResults(cost=c, runtime_s=runtime, path=merge_contraction_paths(len(tn), paths),
                                      disconnected_costs=list(dc), disconnected_paths=paths)
                   for c, _gid, dc, paths in merged]
        return self._dump_results(tn, results)
