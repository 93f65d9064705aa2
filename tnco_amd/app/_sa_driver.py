"""The SA driver shared by method='sa' with and without a width bound.

Host mirror of /root/reference/tnco/app/infinite_memory/sa.py:100-257 and
tnco/app/finite_width/sa.py:117-289.  What the reference does per run in a loky worker process
(random initial path -> ContractionTree -> Optimizer -> `for beta in betas: opt.update(prob)`) is
done here for ALL runs at once on the GPU: one BatchedOptimizer per connected component, one
kernel launch per chunk of the beta schedule.  With torch.distributed initialised, the run list is
sharded over the ranks (one GPU each, tnco_amd/parallel.py) and merged with one all-reduce(min) +
one all-gather of the heads.
"""
from __future__ import annotations

import os
import sys
from time import perf_counter
from typing import Iterable

import numpy as np

from .. import core, parallel
from ..ctree import get_contraction, pack_masks, ssa_to_linear, unpack_mask
from .app import cost_to_decimal
from .tn import get_connected_components

__all__ = ["run_sa", "merge_contraction_paths", "split_contraction_path", "expand_betas"]


def _pair_lists(a: np.ndarray) -> list:
    """[k, steps, 2] -> k lists of `steps` (x, y) tuples.  (One flat tolist() + zip: the nested tolist() of
    1 024 paths of 511 steps built half a million two-element lists first -- 0.2 s per call.)"""
    k, steps, _two = a.shape
    flat = np.ascontiguousarray(a).ravel().tolist()
    pairs = list(zip(flat[0::2], flat[1::2]))
    return [pairs[j * steps:(j + 1) * steps] for j in range(k)]


class _LivePositions:
    """Positions of the live tensors of a linear (einsum) path: a Fenwick tree over creation order.

    A linear path names a tensor by its position in a list from which contracted tensors are removed and to
    which results are appended.  Creation order never changes, so "position -> tensor" is a rank query and
    "tensor -> position" a prefix count over alive flags: O(log n) each instead of list.index + list.pop."""

    def __init__(self, n_initial: int, n_total: int):
        self.size = 1
        while self.size < max(n_total, 1):
            self.size *= 2
        self.tree = [0] * (self.size + 1)
        for slot in range(1, self.size + 1):  # O(n) build: the first n_initial slots alive
            self.tree[slot] += 1 if slot <= n_initial else 0
            up = slot + (slot & -slot)
            if up <= self.size:
                self.tree[up] += self.tree[slot]
        self.alive = [True] * n_initial + [False] * (n_total - n_initial)
        self.n_alive = n_initial
        self.n_slots = n_initial

    def _add(self, slot: int, delta: int) -> None:
        slot += 1
        while slot <= self.size:
            self.tree[slot] += delta
            slot += slot & -slot

    def slot_at(self, position: int) -> int:
        """The creation slot of the live tensor at `position`."""
        if not 0 <= position < self.n_alive:
            raise IndexError(position)
        at, step, remaining = 0, self.size, position + 1
        while step:
            if at + step <= self.size and self.tree[at + step] < remaining:
                at += step
                remaining -= self.tree[at]
            step //= 2
        return at

    def position_of(self, slot: int) -> int:
        """The position of live slot `slot` (number of live slots before it)."""
        if not (0 <= slot < self.n_slots and self.alive[slot]):
            raise KeyError(slot)
        count, at = 0, slot
        while at:
            count += self.tree[at]
            at -= at & -at
        return count

    def remove(self, slot: int) -> None:
        self.alive[slot] = False
        self.n_alive -= 1
        self._add(slot, -1)

    def append(self) -> int:
        slot = self.n_slots
        self.alive[slot] = True
        self.n_slots += 1
        self.n_alive += 1
        self._add(slot, 1)
        return slot


def merge_contraction_paths(n_tensors: int, paths: Iterable[list], *, autocomplete: bool = True) -> list:
    """Several linear paths over the same `n_tensors` tensors, each touching tensors of its own -> ONE linear path
    that runs them one after the other (contract: tnco/utils/tn.py:334-361; `autocomplete` appends the (0, 1)
    steps that join what is left).  A path that contracts a tensor another path consumed is refused.

    Built as: every path replayed over its own position structure to learn WHICH tensors a step joins (the
    originals keep their slots; a result is tied to the slot the merged structure gives it), the same step then
    read back as positions of the merged structure."""
    paths = [list(p) for p in paths]
    n_steps = sum(len(p) for p in paths)
    merged = _LivePositions(n_tensors, n_tensors + n_steps)
    out = []
    for path in paths:
        own = _LivePositions(n_tensors, n_tensors + len(path))
        to_merged = {}  # slots of this path's results -> their slots in the merged structure
        for step in path:
            try:
                slots = [own.slot_at(int(q)) for q in step]
                if len(slots) != 2 or slots[0] == slots[1]:
                    raise IndexError(step)
                where = sorted(merged.position_of(to_merged.get(s, s)) for s in slots)
            except (IndexError, KeyError, TypeError) as e:
                raise ValueError("'paths' are not valid or not disconnected.") from e
            for s in slots:
                own.remove(s)
                merged.remove(to_merged.get(s, s))
            to_merged[own.append()] = merged.append()
            out.append((where[0], where[1]))
    if autocomplete:
        out.extend([(0, 1)] * (merged.n_alive - 1))
    return out


def split_contraction_path(n_tensors: int, path: Iterable[tuple[int, int]]) -> list:
    """tnco/utils/tn.py:404-517 (default options): a linear path over all tensors -> one path per
    connected component (the inverse of merge_contraction_paths), each still indexed over all tensors."""
    path = [tuple(sorted(p)) for p in path]
    tensors = list(range(n_tensors))
    comp = list(range(n_tensors + len(path)))  # union-find over tensor ids (intermediates: n_tensors + step)

    def find(a):
        while comp[a] != a:
            comp[a] = comp[comp[a]]
            a = comp[a]
        return a

    ids = []
    for i, (x, y) in enumerate(path):
        ty, tx = tensors.pop(y), tensors.pop(x)
        z = n_tensors + i
        tensors.append(z)
        ids.append((tx, ty, z))
        for t in (tx, ty):
            ra, rb = find(t), find(z)
            comp[max(ra, rb)] = min(ra, rb)
    order = list(dict.fromkeys(find(t) for t in range(n_tensors) if any(find(t) == find(z) for _x, _y, z in ids)))
    paths = {c: [] for c in order}
    pos = {c: list(range(n_tensors)) for c in order}
    for tx, ty, z in ids:
        c = find(z)
        x, y = sorted((pos[c].index(tx), pos[c].index(ty)))
        paths[c].append((x, y))
        pos[c].pop(y)
        pos[c].pop(x)
        pos[c].append(z)
    return [paths[c] for c in order if paths[c]]


def _positive_int(n_steps):
    try:
        whole = int(n_steps)
    except (TypeError, ValueError):
        whole = None
    if whole is None or whole != n_steps or whole <= 0:
        raise ValueError("'n_steps' must be a positive number.")
    return whole


def expand_betas(betas, n_steps):
    """The inverse temperatures of a call as one float64 array (contract and messages: sa.py:141-156, 201).

    A pair (beta_ini, beta_end) is the arithmetic schedule the reference takes from
    more_itertools.numeric_range(b0, b1, step), step = (b1 - b0) / n_steps computed once: the values
    b0 + k * step, k = 0, 1, ... that have not reached b1 -- of which the step loop uses at most n_steps.  Anything
    else is a list of betas, cut at n_steps when that is given."""
    limit = None if n_steps is None else _positive_int(n_steps)
    if isinstance(betas, tuple) and len(betas) == 2:
        if limit is None:
            raise ValueError("'n_steps' must be provided if 'betas' has the format '(beta_min, beta_max)'.")
        first, last = betas
        if first == last:
            raise ValueError("'betas' must use the format '(beta_ini, beta_end)', with 'beta_ini != beta_end'.")
        step = (last - first) / limit
        ramp = float(first) + np.arange(limit, dtype=np.float64) * float(step)  # (same IEEE product and sum as b0 + k * step)
        reached = (ramp >= last) if step > 0 else (ramp <= last)
        return ramp[:int(np.argmax(reached))] if reached.any() else ramp
    schedule = np.asarray([float(b) for b in betas], np.float64)
    return schedule if limit is None else schedule[:limit]


def replica_seeds(rng, k: int) -> list:
    """`rng.choices(range(2**32), k=k)` (tnco/app/infinite_memory/sa.py:237) -- for large k through numpy: CPython's
    choices() is [floor(random() * n) for _ in range(k)] (Lib/random.py), random() the 53-bit double of two
    MT19937 outputs, which is what numpy's legacy RandomState.random_sample makes of the same state; the
    generator is left where choices() would leave it.  65 536 seeds: 12 ms -> 1 ms."""
    if k < 4096 or type(rng).random is not __import__("random").Random.random:
        return rng.choices(range(2**32), k=k)
    ver, state, gauss = rng.getstate()
    rs = np.random.RandomState()
    rs.set_state(("MT19937", np.asarray(state[:624], np.uint32), int(state[624])))
    out = np.floor(rs.random_sample(k) * 4294967296.0).astype(np.int64).tolist()
    _, key, pos = rs.get_state()[:3]
    rng.setstate((ver, tuple(int(x) for x in key) + (int(pos),), gauss))
    return out


class _Component:
    """One connected component flattened to bit positions."""

    def __init__(self, tn, cc):
        self.tensors = tuple(cc)
        all_inds = tn.ts_inds  # (a property that rebuilds the tuple: once, not once per tensor)
        ts = [all_inds[t] for t in cc]
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

    def names(self, mask) -> frozenset:
        return frozenset(self.inds_order[p] for p in unpack_mask(mask))


# cost_to_decimal prints 6 significant digits ('%g'): two raw totals whose Decimals tie or swap differ by at
# most this relative amount (half a unit of the 6th digit per component, doubled for safety)
DECIMAL_TIE_MARGIN = 2e-5
PROGRESS_POINTS = 20  # best-cost reports per component and schedule when `verbose` / `progress` ask for them


def run_sa(opt, tn, betas, n_steps, n_runs, n_projs, timeout, *, top_k, sweeps_per_launch, prob, device,
           update_slices: int | None, initial_trees: str = "greedy", progress=None):
    """Returns (merged, runtime) with merged = [(cost, global run id, per-component costs, per-component
    paths, per-component slices | None, merged path)] sorted, the `top_k` best runs over all ranks.

    Progress (replaces the `status` / `log2_total_cost` shared-memory buffers the reference's worker
    processes write every step and its rich progress bar renders, tnco/parallel.py:229-317,
    tnco/app/infinite_memory/sa.py:208-209): with `opt.verbose` or a `progress` callable, the best
    min_total_cost over this rank's runs is read back (k-select of one on the device) about
    PROGRESS_POINTS times per component and schedule -- printed to stderr when verbose, passed as
    progress(component, sweeps_done, sweeps_total, best_cost), and kept in tn.tags['progress']."""
    finite = update_slices is not None
    if initial_trees not in ("greedy", "kruskal"):
        raise ValueError("'initial_trees' must be 'greedy' or 'kruskal'.")
    betas = expand_betas(betas, n_steps)
    n_runs = int(n_runs)
    if n_runs <= 0:
        raise ValueError("'n_runs' must be a positive number.")
    if tn.sparse_inds and not n_projs:
        raise ValueError("'n_projs' must be provided if 'tn' has sparse indices.")
    seeds = replica_seeds(opt._rng, n_runs)  # sa.py:237

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
    # outputs of Random(seed) consumed so far by every run: the reference shares one generator over
    # the components of a run (tnco/utils/tn.py:163,192)
    draws = np.zeros(n_local, np.uint64)
    series = []
    n_holders = {}
    for xs in tn.ts_inds:
        for i in xs:
            n_holders[i] = n_holders.get(i, 0) + 1
    for ci, comp in enumerate(comps):
        if len(comp.tensors) <= 1 or n_local == 0:  # `if not path`, sa.py:179-183
            handles.append(None)
            continue
        if initial_trees == "greedy":
            # tn.py:175-178: an output index held by two or more tensors is a contractible edge
            keep = [k for k, x in enumerate(comp.inds_order) if x in tn.output_inds and n_holders.get(x, 0) <= 1]
            # large batches are drawn on the GPU (csrc/greedy_device.hip: the same trees, tests/test_gpu_greedy.py);
            # TNCO_HIP_GREEDY=host|device overrides the size rule
            where = os.environ.get("TNCO_HIP_GREEDY", "auto")
            on_gpu = where == "device" or (where == "auto" and n_local * len(comp.tensors) >= 1_000_000)
            links = core.greedy_trees(comp.leaf_positions, comp.n_inds, my_seeds,
                                      output_mask=pack_masks([keep], comp.n_inds)[0], draws=draws,
                                      device=device if on_gpu else None, keep_on_device=on_gpu)
        else:
            links = core.random_trees(comp.leaf_positions, comp.n_inds, my_seeds)
        kw = dict(max_width=opt.max_width, width_type=opt.width_type) if finite else {}
        h = core.BatchedOptimizer(comp.leaf_masks, links, my_seeds, n_inds=comp.n_inds, dims=comp.dims,
                                  output_mask=comp.output_mask, sparse_mask=comp.sparse_mask,
                                  n_projs=n_projs, cost_type=opt.cost_type, device=device, **kw)
        if ci == len(comps) - 1 or all(len(c.tensors) <= 1 for c in comps[ci + 1:]):
            core.greedy_release()  # the last batch of device-drawn trees has been consumed: return its memory
        # sweeps per kernel launch (launches queue up on the device; a launch is also the grain of `timeout` and of the
        # progress series).  Default: 100; 1000 for a small batch or a handle whose trees stay in LDS during a launch
        # (csrc/sa_small.h) -- with few replicas per wavefront a launch ends when the slowest replica ends (100 sweeps: ~10 %
        # of idle tail, 1000: ~3 %), and 1000 sweeps of LDS-resident trees are 10-30 ms.  A small batch in the HBM kernel
        # (spread over the wavefronts) advances at ~4e5 move-evals/s per run, a sweep is ~ n_leaves moves: the launch is
        # sized to ~0.25 s so that `timeout` and the progress series keep their grain on large trees.
        if sweeps_per_launch is not None:
            spl = sweeps_per_launch
        elif finite:
            spl = 100
        elif h.launch_groups == 0:
            spl = 1000
        elif n_local <= 4096:
            spl = min(1000, max(100, int(1e5 / max(1, len(comp.tensors)))))
        else:
            spl = 100
        spl = max(1, int(spl))
        starts = list(range(0, len(betas), spl))
        report_every = max(1, -(-len(starts) // PROGRESS_POINTS)) if (opt.verbose or progress is not None) else 0
        for k, s in enumerate(starts):
            if timeout is not None and perf_counter() - t0 > timeout:
                timed_out = True
                break
            h.run(betas[s:s + spl], prob, update_slices_every=update_slices or 0)
            if timeout is not None:
                h.sync()
            if report_every and ((k + 1) % report_every == 0 or k == len(starts) - 1):
                done = min(len(betas), s + spl)
                best_now = float(h.best(1)[0][0])
                series.append(dict(component=ci, sweeps=done, of=len(betas), best_cost=best_now))
                if progress is not None:
                    progress(ci, done, len(betas), best_now)
                if opt.verbose:
                    print(f"[tnco_amd] rank {rank} component {ci + 1}/{len(comps)}: {done}/{len(betas)} sweeps, "
                          f"{n_local} runs, log2(min_total_cost) = {np.log2(best_now):.4f}", file=sys.stderr, flush=True)
        raw_cost[:, ci] = h.costs()[1]
        handles.append(h)
    runtime = perf_counter() - t0

    # cost of a run = sum of the per-component Decimals (sa.py:215-218), which carry 6 significant
    # digits: the head of `sorted(results)` (sa.py:257) by (Decimal total, run id) can only hold runs
    # whose raw total is within DECIMAL_TIE_MARGIN of the top_k-th smallest raw total, so only those few are
    # turned into Decimals and sorted exactly
    live = [ci for ci, h in enumerate(handles) if h is not None]
    raw_total = raw_cost.sum(axis=1)
    cand = np.arange(n_local)
    if n_local > top_k and np.all(np.isfinite(raw_total)):
        kth = np.partition(raw_total, top_k - 1)[top_k - 1]
        few = np.nonzero(raw_total <= kth * (1 + DECIMAL_TIE_MARGIN))[0]
        if len(few) >= top_k:  # (else, e.g. non-positive costs: sort them all, like the reference's sorted(results))
            cand = few
    dec = {int(r): [cost_to_decimal(raw_cost[r, ci]) if handles[ci] is not None else 0 for ci in range(len(comps))]
           for r in cand}
    totals = {r: sum(d) for r, d in dec.items()}
    order = sorted(dec, key=lambda r: (totals[r], lo + r))[:top_k]
    # best trees of the head: one device pass + one copy back per component (materialised from the
    # checkpoint + rotation log, post-order get_contraction on the device), then path() natively
    paths_by_comp, slices_by_comp, cons = {}, {}, []
    for ci in live:
        comp, h = comps[ci], handles[ci]
        _links, con = h.trees(order, which_min=True)
        cons.append(con)
        paths_by_comp[ci] = _pair_lists(core.linear_paths(con, comp.tensors, len(tn)))
        if finite:
            slices_by_comp[ci] = h.slices_many(order)[1]
    # merge_contraction_paths (tn.py:334-401) of every result, natively and for all of them at once
    if live and order:
        merged_all = _pair_lists(core.merged_paths(cons, [comps[ci].tensors for ci in live], len(tn)))
    else:
        merged_all = [merge_contraction_paths(len(tn), [[] for _ in comps])] * len(order)
    local = []
    for j, r in enumerate(order):
        paths, slices = [], []
        for ci, comp in enumerate(comps):
            if handles[ci] is None:
                paths.append([])
                slices.append(())
            else:
                paths.append(paths_by_comp[ci][j])
                # (bit positions, not the user's index names -- any hashable: only plain data crosses ranks)
                slices.append(tuple(unpack_mask(slices_by_comp[ci][j])) if finite else ())
        local.append((totals[r], lo + r, dec[r], paths, slices, list(merged_all[j])))
    best_raw = float(raw_total.min()) if n_local else float("inf")
    for h in handles:
        if h is not None:
            h.close()

    merged = parallel.merge_heads(local, top_k, rank, world)
    # positions -> names on this side of the exchange: `comps` is the same on every rank
    merged = [(c, gid, dc, paths, [frozenset(comp.inds_order[p] for p in pos) for comp, pos in zip(comps, sl)], mp)
              for c, gid, dc, paths, sl, mp in merged]
    tn.tags["best_raw_cost"] = parallel.global_best(best_raw, rank, world, device)
    tn.tags["n_runs"] = n_runs
    tn.tags["timed_out"] = timed_out
    if series:
        tn.tags["progress"] = series
    return merged, runtime
