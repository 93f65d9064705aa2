"""Batched optimizer handle: the build's counterpart of
tnco_core.optimize.infinite_memory.Optimizer_<cost>
(/root/reference/include/tnco/optimize/infinite_memory/optimizer.hpp:262-310) for
MANY replicas on one GPU, through the C ABI of include/tnco_hip.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from .ctree import n_words

__all__ = ["BatchedOptimizer", "DeviceLinks", "random_trees", "greedy_trees", "linear_paths", "merged_paths", "PROB_BASE", "PROB_GREEDY", "PROB_MH"]

PROB_BASE, PROB_GREEDY, PROB_MH = _lib.PROB_BASE, _lib.PROB_GREEDY, _lib.PROB_MH
_PROB = {"base": PROB_BASE, "greedy": PROB_GREEDY, "mh": PROB_MH,
         "metropolishastings": PROB_MH, 0: 0, 1: 1, 2: 2}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def holders_csr(leaf_positions: Sequence[Sequence[int]], n_inds: int):
    """leaf -> index positions  ==>  CSR index -> ascending leaves."""
    cnt = np.zeros(n_inds + 1, np.int32)
    for xs in leaf_positions:
        for p in xs:
            cnt[p + 1] += 1
    off = np.cumsum(cnt, dtype=np.int32)
    fill = off[:-1].copy()
    hold = np.zeros(int(off[-1]), np.int32)
    for t, xs in enumerate(leaf_positions):
        for p in xs:
            hold[fill[p]] = t
            fill[p] += 1
    return off, hold


def random_trees(leaf_positions, n_inds: int, seeds, n_threads: int = 0) -> np.ndarray:
    """Seeded random initial trees for one connected component: links[R, 3, N].

    Native batched twin of ctree.random_contraction (csrc/host_trees.cpp).
    """
    L = _lib.load()
    n = len(leaf_positions)
    off, hold = holders_csr(leaf_positions, n_inds)
    seeds = np.ascontiguousarray(np.asarray(seeds, np.uint64) & np.uint64(0xFFFFFFFF), np.uint32)
    out = np.empty((len(seeds), 3, 2 * n - 1), np.int32)
    rc = L.tnco_hip_random_trees(n, n_inds, _ptr(off), _ptr(hold), len(seeds), _ptr(seeds), _ptr(out), n_threads)
    if rc:
        raise ValueError("tensor network component is not connected.")
    return out


class DeviceLinks:
    """Initial trees [R, 3, N] left in the memory of GPU `device` by greedy_trees(..., keep_on_device=True):
    BatchedOptimizer takes them as `links` without a trip through the host.  The memory belongs to the
    library and is re-used by the next greedy_trees call on a device: build the optimizer first."""

    def __init__(self, ptr: int, shape, device: int):
        self.ptr, self.shape, self.device = int(ptr), tuple(shape), int(device)

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, np.int32)
        _lib.check(_lib.load().tnco_hip_copy_to_host(_ptr(out), C.c_void_p(self.ptr), out.nbytes))
        return out


def greedy_trees(leaf_positions, n_inds: int, seeds, output_mask=None, draws=None, n_threads: int = 0,
                 device: int | None = None, keep_on_device: bool = False):
    """Initial trees as the reference draws them (Random(seed).shuffle + opt_einsum's greedy,
    tnco/utils/tn.py:189-230) for one connected component: links[R, 3, N].

    Native batched twin of ctree.greedy_contraction: on host threads (csrc/host_greedy.cpp) or, with
    `device` = a GPU ordinal, on that GPU (csrc/greedy_device.hip: one wavefront per tree; networks
    outside the kernel's limits go to the host version inside the call) -- the same trees either way.
    `draws` (uint64[R], updated in place): outputs of Random(seed) consumed by the components before
    this one.  `keep_on_device` (with `device`): return a DeviceLinks instead of an array."""
    L = _lib.load()
    n = len(leaf_positions)
    off, hold = holders_csr(leaf_positions, n_inds)
    seeds = np.ascontiguousarray(np.asarray(seeds, np.uint64) & np.uint64(0xFFFFFFFF), np.uint32)
    om = None if output_mask is None else np.ascontiguousarray(output_mask, np.uint64)
    if draws is not None and (draws.dtype != np.uint64 or draws.shape != (len(seeds),) or not draws.flags.c_contiguous):
        raise ValueError("'draws' must be a contiguous uint64 array with one entry per seed.")
    shape = (len(seeds), 3, 2 * n - 1)
    if keep_on_device and device is None:
        raise ValueError("'keep_on_device' needs a 'device'.")
    out = None if keep_on_device else np.empty(shape, np.int32)
    dptr = C.c_void_p()
    if device is None:
        rc = L.tnco_hip_greedy_trees(n, n_inds, _ptr(off), _ptr(hold), _ptr(om), len(seeds), _ptr(seeds),
                                     _ptr(draws), _ptr(out), n_threads)
    else:
        rc = L.tnco_hip_greedy_trees_device(int(device), n, n_inds, _ptr(off), _ptr(hold), _ptr(om), len(seeds),
                                            _ptr(seeds), _ptr(draws), _ptr(out),
                                            C.byref(dptr) if keep_on_device else None, n_threads)
    if rc:
        raise ValueError("greedy initial contraction failed (component not connected?).")
    return DeviceLinks(dptr.value, shape, device) if keep_on_device else out


def release_cached() -> int:
    """Gives back the device memory the library keeps from destroyed optimizers for the next one
    (csrc/dev_cache.h; `TNCO_HIP_CACHE_MB`); returns the bytes that were held."""
    L = _lib.load()
    held = int(L.tnco_hip_diag_cached_bytes())
    L.tnco_hip_release_cached()
    return held


def greedy_release() -> None:
    """Frees the device memory greedy_trees(..., device=) keeps between calls (one block per process,
    ~2.5 GB after 65536 x 512-leaf trees); a DeviceLinks handed out earlier is invalid afterwards."""
    _lib.load().tnco_hip_greedy_device_release()


def linear_paths(contraction, tensors_pos, n_tensors: int, n_threads: int = 0) -> np.ndarray:
    """ContractionTree.path() (tnco/ctree.py:350-388) of k contractions [k, nc-1, 3] -> [k, nc-1, 2]."""
    L = _lib.load()
    con = np.ascontiguousarray(contraction, np.int32)
    tp = np.ascontiguousarray(tensors_pos, np.int32)
    k, steps, _ = con.shape
    if steps != len(tp) - 1:
        raise ValueError("'contraction' does not match 'tensors_pos'.")
    out = np.empty((k, steps, 2), np.int32)
    if L.tnco_hip_linear_paths(int(n_tensors), len(tp), _ptr(tp), k, _ptr(con), _ptr(out), n_threads):
        raise ValueError("'contraction' is not valid.")
    return out


def merged_paths(contractions, tensors_pos, n_tensors: int, *, autocomplete: bool = True, n_threads: int = 0):
    """merge_contraction_paths (tnco/utils/tn.py:334-401) for k results at once.

    contractions: one array [k, nc_i - 1, 3] per connected component (get_contraction triples with
    component-local node ids, as BatchedOptimizer.trees returns them); tensors_pos: per component the
    ascending positions of its tensors among all n_tensors.  Returns [k, steps, 2] with steps =
    sum(nc_i - 1) (+ the pairs (0, 1) that join the components when `autocomplete`)."""
    L = _lib.load()
    k = len(contractions[0]) if contractions else 0
    parts, off = [], 0
    for con, tp in zip(contractions, tensors_pos):
        con = np.asarray(con, np.int64)
        tp = np.asarray(tp, np.int64)
        nc = len(tp)
        steps = nc - 1
        inv = np.empty((k, steps), np.int64)  # node id - nc -> step at which it is created
        np.put_along_axis(inv, con[:, :, 2] - nc, np.broadcast_to(np.arange(steps), (k, steps)), axis=1)
        g = np.empty((k, steps, 3), np.int64)
        for j in range(2):
            x = con[:, :, j]
            leaf = x < nc
            g[:, :, j] = np.where(leaf, tp[np.where(leaf, x, 0)],
                                  n_tensors + off + np.take_along_axis(inv, np.where(leaf, 0, x - nc), axis=1))
        g[:, :, 2] = n_tensors + off + np.arange(steps)
        parts.append(g)
        off += steps
    tri = np.ascontiguousarray(np.concatenate(parts, axis=1) if parts else np.zeros((k, 0, 3)), np.int32)
    out = np.empty((k, off, 2), np.int32)
    if L.tnco_hip_linear_paths_ssa(int(n_tensors), off, k, _ptr(tri), _ptr(out), n_threads):
        raise ValueError("'paths' are not valid or not disconnected.")
    if autocomplete:
        extra = n_tensors - off - 1
        if extra > 0:
            tail = np.broadcast_to(np.array([0, 1], np.int32), (k, extra, 2))
            out = np.concatenate([out, tail], axis=1)
    return out


class BatchedOptimizer:
    """R independent SA replicas of one tensor network on one GPU.

    Args:
        leaf_masks: uint64 [n_leaves, W] legs of the input tensors.
        links: int32 [R, 3, N] (left, right, parent per replica) or [3, N]
            (one tree shared by all replicas).
        seeds: R mt19937 seeds (taken mod 2**32 like prng.seed(size_t),
            include/tnco/optimize/optimizer.hpp:73).
        dims: int or per-index sequence.  output_mask / sparse_mask: uint64 [W].
        node_masks: optional explicit legs of every node, [R, N, W] or [N, W].
    """

    def __init__(self, leaf_masks, links, seeds, *, n_inds: int, dims=2, output_mask=None,
                 sparse_mask=None, n_projs: int | None = None, cost_type: str = "float64",
                 disable_shared_inds: bool = False, node_masks=None, device: int = 0,
                 max_width: float | None = None, width_type: str = "float32",
                 max_number_new_slices: int = 0, skip_slices=None, slices=None, min_links=None,
                 min_slices=None, prng_states=None, steps_done: int = 0):
        """Finite width (max_width finite): the batched counterpart of
        finite_width.greedy.Optimizer_<cost>_<width>
        (include/tnco/optimize/finite_width/greedy/optimizer.hpp:462-518).

        Restoring a batch (what Optimizer.__reduce__ round-trips in the reference,
        tnco/optimize/infinite_memory/optimizer.py:243-245, finite_width/optimizer.py:343-346):
        `links` = the current trees, `min_links` [R, 3, N] or [3, N] = min_ctree, `slices` /
        `min_slices` [R, W] or [W], `prng_states` [R, 625] = prng_state of every replica (then
        `seeds` may be None), `steps_done` = sweeps already run (the n of `n % update_slices`).
        As in the reference the caches are rebuilt and min_total_cost = get_cost(min_ctree)."""
        self._h = None
        L = _lib.load()
        if cost_type not in ("float64", "float32"):
            raise NotImplementedError(f"cost_type={cost_type!r} is not supported on the GPU path "
                                      "(float64 / float32 only).")
        leaf_masks = np.ascontiguousarray(leaf_masks, np.uint64)
        n = leaf_masks.shape[0]
        W = n_words(n_inds)
        if leaf_masks.shape != (n, W):
            raise ValueError("'leaf_masks' has the wrong shape.")
        N = 2 * n - 1
        on_device = isinstance(links, DeviceLinks)
        if on_device and links.device != int(device):
            raise ValueError("'links' are in the memory of another device.")
        if not on_device:
            links = np.ascontiguousarray(links, np.int32)
        ps = None
        if prng_states is not None:
            ps = np.ascontiguousarray(prng_states, np.uint32)
            if ps.ndim != 2 or ps.shape[1] != 625 or (seeds is not None and len(seeds) != len(ps)):
                raise ValueError("'prng_states' must be [n_replicas, 625].")
        if seeds is None:
            if ps is None:
                raise ValueError("'seeds' or 'prng_states' is needed.")
            seeds = np.zeros(len(ps), np.uint32)
        seeds = np.ascontiguousarray(np.asarray(seeds, np.uint64) & np.uint64(0xFFFFFFFF), np.uint32)
        R = len(seeds)
        if links.shape == (3, N):
            stride = 0
        elif links.shape == (R, 3, N):
            stride = 3 * N
        else:
            raise ValueError("'links' has the wrong shape.")
        d = _lib.Desc()
        d.n_leaves, d.n_inds, d.n_replicas = n, n_inds, R
        d.leaf_masks = _ptr(leaf_masks)
        om = None if output_mask is None else np.ascontiguousarray(output_mask, np.uint64)
        d.output_mask = _ptr(om)
        d.links, d.links_stride = (C.c_void_p(links.ptr) if on_device else _ptr(links)), stride
        nm = None
        if node_masks is not None:
            nm = np.ascontiguousarray(node_masks, np.uint64)
            if nm.shape == (N, W):
                d.node_masks_stride = 0
            elif nm.shape == (R, N, W):
                d.node_masks_stride = N * W
            else:
                raise ValueError("'node_masks' has the wrong shape.")
            d.node_masks = _ptr(nm)
        dv = None
        if np.ndim(dims) == 0:
            if int(dims) != dims or int(dims) <= 0:
                raise ValueError("Dimensions must be positive numbers")
            d.dim_uniform = int(dims)
        else:
            dv = np.ascontiguousarray(dims, np.uint64)
            if dv.shape != (n_inds,):
                raise ValueError("Wrong number of dimensions.")
            d.dims = _ptr(dv)
        sm = None
        if sparse_mask is not None and np.any(np.asarray(sparse_mask)):
            sm = np.ascontiguousarray(sparse_mask, np.uint64)
            d.sparse_mask = _ptr(sm)
            if not n_projs or n_projs <= 0:
                raise RuntimeError("'n_projs' must be a positive number.")
            d.n_projs = int(n_projs)
        d.cost_dtype = _lib.F64 if cost_type == "float64" else _lib.F32
        d.disable_shared_inds = int(bool(disable_shared_inds))
        d.seeds = _ptr(seeds)
        d.device = int(device)
        self.finite_width = max_width is not None and max_width < float("inf")
        d.max_width = float(max_width) if self.finite_width else float("nan")
        if self.finite_width and width_type not in ("float32", "float64"):
            raise NotImplementedError(f"width_type={width_type!r} is not supported on the GPU path "
                                      "(float32, float64).")
        d.width_dtype = _lib.F64 if width_type == "float64" else _lib.F32
        d.max_number_new_slices = int(max_number_new_slices)
        sk = None if skip_slices is None else np.ascontiguousarray(skip_slices, np.uint64)
        sl = None if slices is None else np.ascontiguousarray(slices, np.uint64)
        msl = None if min_slices is None else np.ascontiguousarray(min_slices, np.uint64)
        for x in (sl, msl):
            if x is not None and x.shape not in ((W,), (R, W)):
                raise ValueError("'slices' / 'min_slices' must be [W] or [n_replicas, W].")
        if sl is not None and msl is not None and sl.shape != msl.shape:
            raise ValueError("'slices' and 'min_slices' must have the same shape.")
        per_replica = any(x is not None and x.ndim == 2 for x in (sl, msl))
        d.skip_slices, d.slices, d.min_slices = _ptr(sk), _ptr(sl), _ptr(msl)
        d.slices_stride = W if per_replica else 0
        ml = None
        if min_links is not None:
            ml = np.ascontiguousarray(min_links, np.int32)
            if ml.shape == (3, N):
                d.min_links_stride = 0
            elif ml.shape == (R, 3, N):
                d.min_links_stride = 3 * N
            else:
                raise ValueError("'min_links' has the wrong shape.")
            d.min_links = _ptr(ml)
        d.prng_states = _ptr(ps)
        self._steps_done = int(steps_done)
        h = C.c_void_p()
        _lib.check(L.tnco_hip_create(C.byref(d), C.byref(h)))
        self._h = h
        self._L = L
        self.n_leaves, self.n_nodes, self.n_inds, self.n_words, self.n_replicas = n, N, n_inds, W, R
        self.cost_type = cost_type
        self.device = int(device)

    # -- lifetime -----------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.tnco_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the hot path ---------------------------------------------------------
    def run(self, betas, prob="mh", sync: bool = False, update_slices_every: int = 10) -> None:
        """len(betas) sweeps per replica (one `update(prob)` per beta).

        Finite width: sweep number n (counted over all calls) re-slices when
        n % update_slices_every == 0, as tnco/app/finite_width/sa.py:228 drives it."""
        betas = np.ascontiguousarray(betas, np.float64)
        kind = _PROB[prob.lower() if isinstance(prob, str) else prob]
        if self.finite_width:
            _lib.check(self._L.tnco_hip_run_fw(self._h, kind, _ptr(betas), len(betas),
                                               int(update_slices_every), self._steps_done))
        else:
            _lib.check(self._L.tnco_hip_run(self._h, kind, _ptr(betas), len(betas)))
        self._steps_done += len(betas)
        if sync:
            self.sync()

    def slices(self, replica: int):
        """(slices, min_slices) masks of one replica (finite width)."""
        a = np.zeros(self.n_words, np.uint64)
        b = np.zeros(self.n_words, np.uint64)
        _lib.check(self._L.tnco_hip_get_slices(self._h, int(replica), _ptr(a), _ptr(b)))
        return a, b

    def slices_many(self, replicas):
        """(slices, min_slices) masks [k, W] of k replicas (finite width), one copy back."""
        ids = np.ascontiguousarray(replicas, np.int64)
        a = np.zeros((len(ids), self.n_words), np.uint64)
        b = np.zeros((len(ids), self.n_words), np.uint64)
        _lib.check(self._L.tnco_hip_get_slices_many(self._h, len(ids), _ptr(ids), _ptr(a), _ptr(b)))
        return a, b

    def reslice_info(self):
        """(how, n_changed) of every replica's LAST re-slice (diagnostics, tnco_hip_diag_reslice_info): how = 1
        re-priced, 0 rebuilt in full / no slices; n_changed = indices the proposal differed by (-1 unknown)."""
        how = np.empty(self.n_replicas, np.int32)
        nch = np.empty(self.n_replicas, np.int32)
        _lib.check(self._L.tnco_hip_diag_reslice_info(self._h, _ptr(how), _ptr(nch)))
        return how, nch

    def fw_stats(self) -> dict:
        """What this handle's re-slices did since it was created, in replica re-slices (tnco_hip_diag_fw_stats)."""
        o = np.zeros(8, np.int64)
        _lib.check(self._L.tnco_hip_diag_fw_stats(self._h, _ptr(o)))
        return dict(repriced=int(o[0]), fell_back=int(o[1]), too_many_wide=int(o[2]), too_many_changed=int(o[3]),
                    cost_range=int(o[4]), full_rebuild_form=int(o[5]))

    def update(self, beta: float = 0.0, prob="mh") -> None:
        self.run([beta], prob)

    def sync(self) -> None:
        _lib.check(self._L.tnco_hip_sync(self._h))

    def set_stream(self, stream_ptr: int | None) -> None:
        _lib.check(self._L.tnco_hip_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    # -- read-back ------------------------------------------------------------
    def costs(self):
        tot = np.empty(self.n_replicas, np.float64)
        mn = np.empty(self.n_replicas, np.float64)
        _lib.check(self._L.tnco_hip_get_costs(self._h, _ptr(tot), _ptr(mn)))
        return tot, mn

    @property
    def total_cost(self) -> np.ndarray:
        return self.costs()[0]

    @property
    def min_total_cost(self) -> np.ndarray:
        return self.costs()[1]

    def tree(self, replica: int, which_min: bool = False, with_masks: bool = True):
        N, W = self.n_nodes, self.n_words
        l, r, p = (np.empty(N, np.int32) for _ in range(3))
        m = np.empty((N, W), np.uint64) if with_masks else None
        _lib.check(self._L.tnco_hip_get_tree(self._h, int(replica), int(which_min), _ptr(l), _ptr(r), _ptr(p), _ptr(m)))
        return l, r, p, m

    def caches(self, replica: int):
        N, W = self.n_nodes, self.n_words
        cc, pc = np.empty(N, np.float64), np.empty(N, np.float64)
        hy = np.empty((N, W), np.uint64)
        _lib.check(self._L.tnco_hip_get_caches(self._h, int(replica), _ptr(cc), _ptr(pc), _ptr(hy)))
        return cc, pc, hy

    def validate(self, atol: float = 1e-5):
        """(n_bad, first_bad): is_valid(atol) of every replica, recomputed on the device."""
        nb, fb = C.c_int64(0), C.c_int64(-1)
        _lib.check(self._L.tnco_hip_validate(self._h, float(atol), C.byref(nb), C.byref(fb)))
        return nb.value, fb.value

    def is_valid(self, atol: float = 1e-5) -> bool:
        return self.validate(atol)[0] == 0

    def prng_state(self, replica: int) -> np.ndarray:
        out = np.empty(625, np.uint32)
        _lib.check(self._L.tnco_hip_get_prng(self._h, int(replica), _ptr(out)))
        return out

    def prng_state_string(self, replica: int) -> str:
        """The `prng_state` property of the reference (optimize/optimizer.hpp:191-195): the text
        libstdc++ streams for a std::mt19937, 624 state words and the position, space separated."""
        return " ".join(str(int(x)) for x in self.prng_state(replica))

    def set_prng_state(self, replica: int, state625) -> None:
        if isinstance(state625, str):  # a string seed restores the full state (optimizer.hpp:68-71)
            state625 = [int(x) for x in state625.split()]
        st = np.ascontiguousarray(state625, np.uint32)
        if st.shape != (625,):
            raise ValueError("prng state must hold 625 words.")
        _lib.check(self._L.tnco_hip_set_prng(self._h, int(replica), _ptr(st)))

    def prng_states(self, replicas=None) -> np.ndarray:
        """prng_state of k replicas (all when None) in one call: uint32 [k, 625]."""
        if replicas is None:
            out = np.empty((self.n_replicas, 625), np.uint32)
            _lib.check(self._L.tnco_hip_get_prng_many(self._h, self.n_replicas, None, _ptr(out)))
            return out
        ids = np.ascontiguousarray(replicas, np.int64)
        out = np.empty((len(ids), 625), np.uint32)
        _lib.check(self._L.tnco_hip_get_prng_many(self._h, len(ids), _ptr(ids), _ptr(out)))
        return out

    def set_prng_states(self, states, replicas=None) -> None:
        st = np.ascontiguousarray(states, np.uint32)
        ids = None if replicas is None else np.ascontiguousarray(replicas, np.int64)
        k = len(st)
        if st.ndim != 2 or st.shape[1] != 625 or (ids is not None and len(ids) != k) or (ids is None and k > self.n_replicas):
            raise ValueError("prng states must be [k, 625].")
        _lib.check(self._L.tnco_hip_set_prng_many(self._h, k, _ptr(ids), _ptr(st)))

    def snapshot(self) -> dict:
        """Everything Optimizer.__reduce__ round-trips in the reference, for all replicas at once
        (tnco/optimize/infinite_memory/optimizer.py:243-245, finite_width/optimizer.py:343-346): current
        trees, best trees, PRNG states [, slices, min_slices] + the number of sweeps run.  Feed it to
        BatchedOptimizer.restore()."""
        ids = np.arange(self.n_replicas, dtype=np.int64)
        snap = dict(links=self.trees(ids, which_min=False, contraction=False)[0],
                    min_links=self.trees(ids, which_min=True, contraction=False)[0],
                    prng_states=self.prng_states(), steps_done=self._steps_done)
        if self.finite_width:
            snap["slices"], snap["min_slices"] = self.slices_many(ids)
        return snap

    @classmethod
    def restore(cls, snap: dict, leaf_masks, **kw):
        """A new batch from snapshot() + the problem description (leaf_masks and the keyword arguments of the
        constructor).  Like the reference's unpickling it rebuilds the caches from the trees and takes
        min_total_cost = get_cost(min_ctree[, min_slices]) (infinite_memory/optimizer.hpp:61-88)."""
        return cls(leaf_masks, snap["links"], None, min_links=snap["min_links"], prng_states=snap["prng_states"],
                   slices=snap.get("slices"), min_slices=snap.get("min_slices"), steps_done=snap["steps_done"], **kw)

    def best(self, k: int = 1):
        k = int(k)
        c = np.empty(k, np.float64)
        ids = np.empty(k, np.int64)
        _lib.check(self._L.tnco_hip_best(self._h, k, _ptr(c), _ptr(ids)))
        return c, ids

    def trees(self, replicas, which_min: bool = True, contraction: bool = True):
        """links [k, 3, N] (and get_contraction triples [k, n-1, 3], include/tnco/utils.hpp:53-71) of
        k replicas: one device pass, one copy back."""
        ids = np.ascontiguousarray(replicas, np.int64)
        k = len(ids)
        links = np.empty((k, 3, self.n_nodes), np.int32)
        con = np.empty((k, self.n_leaves - 1, 3), np.int32) if contraction else None
        _lib.check(self._L.tnco_hip_get_trees(self._h, k, _ptr(ids), int(which_min), _ptr(links), _ptr(con)))
        return links, con

    def min_cost_to_device(self, device_ptr: int) -> None:
        """min over replicas of min_total_cost, written by the device into device memory."""
        _lib.check(self._L.tnco_hip_min_cost_device(self._h, C.c_void_p(int(device_ptr))))

    def counters(self) -> dict:
        a, b, c, q = (C.c_uint64(0) for _ in range(4))
        _lib.check(self._L.tnco_hip_diag_counters(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(q)))
        f = C.c_uint64(0)
        _lib.check(self._L.tnco_hip_diag_full_copies(self._h, C.byref(f)))
        return dict(moves=a.value, accepted=b.value, improved=c.value, random_picks=q.value,
                    full_copies=f.value)

    def moves_per_replica(self) -> np.ndarray:
        out = np.empty(self.n_replicas, np.uint64)
        _lib.check(self._L.tnco_hip_diag_moves(self._h, _ptr(out)))
        return out

    def kernel_time_ms(self, reset: bool = False):
        """(device ms, schedule chunks) of the run calls since the last reset; with two streams the time from the first
        launch to the end of the last (tnco_hip_diag_kernel_time)."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        _lib.check(self._L.tnco_hip_diag_kernel_time(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def kernel_times_ms(self, reset: bool = False) -> dict:
        """{kernel: (ms, launches)} since the last reset, from HIP events around every launch."""
        ms = np.zeros(4, np.float64)
        n = np.zeros(4, np.int64)
        _lib.check(self._L.tnco_hip_diag_kernel_times(self._h, _ptr(ms), _ptr(n), int(reset)))
        names = ("sa_run_kernel", "fw_move_kernel", "fw_reslice_kernel", "fw_walk_kernel")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    @property
    def launch_groups(self) -> int:
        """Concurrent launches (streams) a step of run() is split into: 1, or 2 when the replicas do not fill
        whole rounds of resident workgroups (tnco_hip_run); 0 for a handle whose sweeps run LDS-resident (small trees;
        larger ones when the whole batch fits the CUs' LDS: csrc/sa_small.h)."""
        return int(self._L.tnco_hip_diag_launch_groups(self._h))

    @property
    def device_bytes(self) -> int:
        return int(self._L.tnco_hip_diag_device_bytes(self._h))
