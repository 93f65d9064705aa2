"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/liboracle.so (the plain-C restatement of the
reference SA path, oracle/tnco_oracle.c) and of oracle/_ref/libref_tree.so (the
real reference Tree compiled from /root/reference, when it was built).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (tnco_amd/) must never import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "liboracle.so"
_REF = _HERE / "_ref" / "libref_tree.so"

PROB_BASE, PROB_GREEDY, PROB_MH = 0, 1, 2


def build(force: bool = False) -> None:
    """Compile the checker (gcc); also _ref when /root/reference is present."""
    if force or not _LIB.exists():
        subprocess.check_call(["make", "-C", str(_HERE), "_build/liboracle.so"],
                              stdout=subprocess.DEVNULL)
    if (force or not _REF.exists()) and Path("/root/reference/include/tnco").is_dir():
        subprocess.call(["make", "-C", str(_HERE), "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB))
        L.orc_mt_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_mt_next.argtypes = [C.c_void_p]
        L.orc_mt_next.restype = C.c_uint32
        L.orc_uniform01.argtypes = [C.c_void_p]
        L.orc_uniform01.restype = C.c_double
        L.orc_uniform_int.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_uniform_int.restype = C.c_uint64
        L.orc_shuffle_i32.argtypes = [C.c_void_p, _i32p, C.c_int64]
        L.orc_traverse.argtypes = [C.c_int32, _i32p, _i32p, _i32p]
        L.orc_traverse.restype = C.c_int
        L.orc_get_contraction.argtypes = [C.c_int32, _i32p, _i32p, _i32p]
        L.orc_get_contraction.restype = C.c_int
        L.orc_swap_with_nn.argtypes = [C.c_int32, _i32p, _i32p, _i32p, C.c_int32]
        L.orc_tree_is_valid.argtypes = [C.c_int32, _i32p, _i32p, _i32p]
        L.orc_tree_is_valid.restype = C.c_int
        L.orc_ctree_is_valid.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _u64p, C.c_int]
        L.orc_ctree_is_valid.restype = C.c_int
        for pf in ("orc_f64_", "orc_f32_"):
            f = getattr(L, pf + "create")
            f.restype = C.c_void_p
            f.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _u64p, C.c_uint64,
                          C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64,
                          C.c_void_p, C.POINTER(C.c_int)]
            f = getattr(L, pf + "create_fw")
            f.restype = C.c_void_p
            f.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _u64p, C.c_uint64,
                          C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64,
                          C.c_void_p, C.c_double, C.c_int, C.c_uint64, C.c_void_p,
                          C.c_void_p, C.POINTER(C.c_int)]
            getattr(L, pf + "destroy").argtypes = [C.c_void_p]
            getattr(L, pf + "set_min").argtypes = [C.c_void_p, _i32p, _i32p, _i32p, _u64p, C.c_void_p]
            getattr(L, pf + "set_min").restype = C.c_int
            getattr(L, pf + "update").argtypes = [C.c_void_p, C.c_int, C.c_double]
            getattr(L, pf + "update_fw").argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
            getattr(L, pf + "run").argtypes = [C.c_void_p, C.c_int, _f64p, C.c_int64]
            getattr(L, pf + "run_fw").argtypes = [C.c_void_p, C.c_int, _f64p, C.c_int64, C.c_int64]
            getattr(L, pf + "is_valid").argtypes = [C.c_void_p, C.c_double]
            getattr(L, pf + "is_valid").restype = C.c_int
            getattr(L, pf + "get_tree").argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_void_p]
            getattr(L, pf + "get_caches").argtypes = [C.c_void_p, _f64p, _f64p, C.c_void_p]
            getattr(L, pf + "total_cost").argtypes = [C.c_void_p]
            getattr(L, pf + "total_cost").restype = C.c_double
            getattr(L, pf + "min_total_cost").argtypes = [C.c_void_p]
            getattr(L, pf + "min_total_cost").restype = C.c_double
            getattr(L, pf + "get_prng").argtypes = [C.c_void_p, _u32p]
            getattr(L, pf + "get_counters").argtypes = [C.c_void_p, _u64p]
            getattr(L, pf + "get_slices_out").argtypes = [C.c_void_p, _u64p, _u64p]
            getattr(L, pf + "get_widths").argtypes = [C.c_void_p, _f64p]
            f = getattr(L, pf + "run_batch")
            f.restype = C.c_double
            f.argtypes = [C.c_int64, C.c_int32, C.c_int32, _i32p, _u64p, C.c_void_p, C.c_uint64, _u32p,
                          C.c_int, _f64p, C.c_int64, C.c_int, _f64p, _f64p, _u64p]
            f = getattr(L, pf + "prob")
            f.restype = C.c_double if pf == "orc_f64_" else C.c_float
            ct = C.c_double if pf == "orc_f64_" else C.c_float
            f.argtypes = [C.c_int, C.c_double, ct, ct]
        _lib = L
    return _lib


def ref_tree():
    """The real reference Tree (None when oracle/_ref was not built)."""
    global _ref
    if _ref is None and _REF.exists():
        R = C.CDLL(str(_REF))
        R.ref_tree_is_valid.argtypes = [C.c_int32, _i32p, _i32p, _i32p]
        R.ref_tree_is_valid.restype = C.c_int
        R.ref_tree_swap_with_nn.argtypes = [C.c_int32, _i32p, _i32p, _i32p, C.c_int32]
        R.ref_tree_swap_with_nn.restype = C.c_int
        R.ref_tree_n_leaves.argtypes = [C.c_int32, _i32p, _i32p, _i32p]
        R.ref_tree_n_leaves.restype = C.c_int
        _ref = R
    return _ref


class MT:
    """std::mt19937 restatement (orc_mt_t)."""

    def __init__(self, seed: int):
        self._buf = (C.c_uint32 * 625)()
        lib().orc_mt_seed(self._buf, int(seed))

    def next(self) -> int:
        return int(lib().orc_mt_next(self._buf))

    def uniform01(self) -> float:
        return float(lib().orc_uniform01(self._buf))

    def uniform_int(self, hi: int) -> int:
        return int(lib().orc_uniform_int(self._buf, int(hi)))

    def shuffle(self, a: np.ndarray) -> None:
        assert a.dtype == np.int32
        lib().orc_shuffle_i32(self._buf, a, len(a))

    def state(self) -> np.ndarray:
        return np.frombuffer(self._buf, dtype=np.uint32).copy()


def traverse(left, right) -> np.ndarray:
    left = np.ascontiguousarray(left, np.int32)
    right = np.ascontiguousarray(right, np.int32)
    out = np.empty(len(left), np.int32)
    k = lib().orc_traverse(len(left), left, right, out)
    return out[:k]


def get_contraction(left, right) -> np.ndarray:
    left = np.ascontiguousarray(left, np.int32)
    right = np.ascontiguousarray(right, np.int32)
    out = np.empty(((len(left) - 1) // 2, 3), np.int32)
    m = lib().orc_get_contraction(len(left), left, right, out.reshape(-1))
    return out[:m]


def _opt_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One SA replica on the CPU restatement (infinite memory or finite width).

    Mirrors tnco_core.optimize.infinite_memory.Optimizer_<cost> /
    finite_width.greedy.Optimizer_<cost>_<width>
    (include/tnco/optimize/infinite_memory/optimizer.hpp:262-310).
    """

    def __init__(self, left, right, parent, inds, *, n_inds, dims=2, sparse=None,
                 n_projs=0, disable_shared_inds=False, seed=0, mt_state=None,
                 cost_type="float64", max_width=None, width_type="float32",
                 max_number_new_slices=0, skip_slices=None, slices=None, min_tree=None, min_slices=None):
        self._pf = {"float64": "orc_f64_", "float32": "orc_f32_"}[cost_type]
        L = lib()
        left = np.ascontiguousarray(left, np.int32)
        right = np.ascontiguousarray(right, np.int32)
        parent = np.ascontiguousarray(parent, np.int32)
        N = len(left)
        n_leaves = (N + 1) // 2
        W = max(1, (n_inds + 63) // 64)
        inds = np.ascontiguousarray(inds, np.uint64).reshape(N, W)
        self.N, self.W, self.n_leaves, self.n_inds = N, W, n_leaves, n_inds
        dims_vec = None
        dim_uniform = 0
        if np.ndim(dims) == 0:
            dim_uniform = int(dims)
        else:
            dims_vec = np.ascontiguousarray(dims, np.uint64)
            assert len(dims_vec) == n_inds
        sp = None if sparse is None else np.ascontiguousarray(sparse, np.uint64)
        mt = None if mt_state is None else np.ascontiguousarray(mt_state, np.uint32)
        st = C.c_int(0)
        self.fw = max_width is not None
        if not self.fw:
            self._h = getattr(L, self._pf + "create")(
                n_leaves, n_inds, left, right, parent, inds.reshape(-1), dim_uniform,
                _opt_ptr(dims_vec), _opt_ptr(sp), int(n_projs), int(disable_shared_inds),
                int(seed), _opt_ptr(mt), C.byref(st))
        else:
            sk = None if skip_slices is None else np.ascontiguousarray(skip_slices, np.uint64)
            sl = None if slices is None else np.ascontiguousarray(slices, np.uint64)
            self._h = getattr(L, self._pf + "create_fw")(
                n_leaves, n_inds, left, right, parent, inds.reshape(-1), dim_uniform,
                _opt_ptr(dims_vec), _opt_ptr(sp), int(n_projs), int(disable_shared_inds),
                int(seed), _opt_ptr(mt), float(max_width), int(width_type == "float32"),
                int(max_number_new_slices), _opt_ptr(sk), _opt_ptr(sl), C.byref(st))
        self.status = st.value
        if not self.status and min_tree is not None:
            # `_min_ctree` / `_min_slices` of the reference's constructors (what __reduce__ round-trips)
            ml, mr, mp, mi = (np.ascontiguousarray(x, t) for x, t in zip(min_tree, (np.int32,) * 3 + (np.uint64,)))
            msl = None if min_slices is None else np.ascontiguousarray(min_slices, np.uint64)
            self.status = int(self._f("set_min")(self._h, ml, mr, mp, mi.reshape(-1), _opt_ptr(msl)))
        if self.status:
            self.close()
            if self.status == 20:
                raise ValueError("Precision is too low.")
            if self.status in (10, 11):
                raise ValueError("Contraction is not valid.")
            raise ValueError(f"Tree is not valid (code {self.status}).")

    def _f(self, name):
        return getattr(lib(), self._pf + name)

    def close(self):
        if getattr(self, "_h", None):
            self._f("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, prob_kind: int, beta: float = 0.0, update_slices: bool = True):
        if self.fw:
            self._f("update_fw")(self._h, prob_kind, float(beta), int(update_slices))
        else:
            self._f("update")(self._h, prob_kind, float(beta))

    def run(self, prob_kind: int, betas, update_slices_every: int = 10):
        betas = np.ascontiguousarray(betas, np.float64)
        if self.fw:
            self._f("run_fw")(self._h, prob_kind, betas, len(betas), int(update_slices_every))
        else:
            self._f("run")(self._h, prob_kind, betas, len(betas))

    def is_valid(self, atol: float = 1e-5) -> int:
        return int(self._f("is_valid")(self._h, float(atol)))

    def tree(self, which_min: bool = False, with_inds: bool = True):
        l = np.empty(self.N, np.int32)
        r = np.empty(self.N, np.int32)
        p = np.empty(self.N, np.int32)
        m = np.empty((self.N, self.W), np.uint64) if with_inds else None
        self._f("get_tree")(self._h, int(which_min), l, r, p, _opt_ptr(m))
        return l, r, p, m

    def caches(self):
        cc = np.empty(self.N, np.float64)
        pc = np.empty(self.N, np.float64)
        hy = np.empty((self.N, self.W), np.uint64)
        self._f("get_caches")(self._h, cc, pc, _opt_ptr(hy))
        return cc, pc, hy

    @property
    def total_cost(self) -> float:
        return float(self._f("total_cost")(self._h))

    @property
    def min_total_cost(self) -> float:
        return float(self._f("min_total_cost")(self._h))

    def prng_state(self) -> np.ndarray:
        out = np.empty(625, np.uint32)
        self._f("get_prng")(self._h, out)
        return out

    def counters(self):
        out = np.zeros(3, np.uint64)
        self._f("get_counters")(self._h, out)
        return dict(moves=int(out[0]), accepted=int(out[1]), improved=int(out[2]))

    def slices(self):
        a = np.zeros(self.W, np.uint64)
        b = np.zeros(self.W, np.uint64)
        self._f("get_slices_out")(self._h, a, b)
        return a, b

    def widths(self):
        w = np.zeros(self.N, np.float64)
        self._f("get_widths")(self._h, w)
        return w


def run_batch(links, leaf_masks, seeds, betas, *, n_inds, dims=2, output_mask=None,
              prob_kind=PROB_MH, n_threads=0, cost_type="float64"):
    """R replicas over OpenMP threads; returns (seconds in the update loops, total, min, moves)."""
    pf = {"float64": "orc_f64_", "float32": "orc_f32_"}[cost_type]
    links = np.ascontiguousarray(links, np.int32)
    R, _, N = links.shape
    leaf_masks = np.ascontiguousarray(leaf_masks, np.uint64)
    seeds = np.ascontiguousarray(np.asarray(seeds, np.uint64) & np.uint64(0xFFFFFFFF), np.uint32)
    betas = np.ascontiguousarray(betas, np.float64)
    om = None if output_mask is None else np.ascontiguousarray(output_mask, np.uint64)
    tot, mn = np.empty(R, np.float64), np.empty(R, np.float64)
    mv = np.zeros(R, np.uint64)
    dt = getattr(lib(), pf + "run_batch")(R, (N + 1) // 2, n_inds, links.reshape(-1), leaf_masks.reshape(-1),
                                           _opt_ptr(om), int(dims), seeds, prob_kind, betas, len(betas),
                                           int(n_threads), tot, mn, mv)
    return float(dt), tot, mn, mv


def prob(kind: int, beta: float, delta: float, old: float, cost_type="float64") -> float:
    pf = {"float64": "orc_f64_", "float32": "orc_f32_"}[cost_type]
    return float(getattr(lib(), pf + "prob")(kind, float(beta), delta, old))
