"""ctypes binding of libtnco_hip.so (the C ABI declared in include/tnco_hip.h).

The shared object is built in-tree by `make -C tnco_amd/csrc` (hipcc,
--offload-arch=gfx950).  There is NO fallback: if it cannot be loaded the
import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import warnings
from pathlib import Path

_PATH = Path(os.environ.get("TNCO_HIP_LIB", Path(__file__).resolve().parent / "libtnco_hip.so"))

OK, EINVAL, ERUNTIME, ENOTIMPL = 0, 1, 2, 3
PROB_BASE, PROB_GREEDY, PROB_MH = 0, 1, 2
F64, F32 = 0, 1


class Desc(C.Structure):
    _fields_ = [
        ("n_leaves", C.c_int32),
        ("n_inds", C.c_int32),
        ("n_replicas", C.c_int64),
        ("leaf_masks", C.c_void_p),
        ("output_mask", C.c_void_p),
        ("links", C.c_void_p),
        ("links_stride", C.c_int64),
        ("node_masks", C.c_void_p),
        ("node_masks_stride", C.c_int64),
        ("dim_uniform", C.c_uint64),
        ("dims", C.c_void_p),
        ("sparse_mask", C.c_void_p),
        ("n_projs", C.c_uint64),
        ("cost_dtype", C.c_int32),
        ("disable_shared_inds", C.c_int32),
        ("seeds", C.c_void_p),
        ("device", C.c_int32),
        ("width_dtype", C.c_int32),
        ("max_width", C.c_double),
        ("max_number_new_slices", C.c_uint64),
        ("skip_slices", C.c_void_p),
        ("slices", C.c_void_p),
        ("min_links", C.c_void_p),
        ("min_links_stride", C.c_int64),
        ("min_slices", C.c_void_p),
        ("slices_stride", C.c_int64),
        ("prng_states", C.c_void_p),
    ]


EXPORTS = [
    "tnco_hip_create", "tnco_hip_run", "tnco_hip_run_fw", "tnco_hip_get_slices", "tnco_hip_get_slices_many", "tnco_hip_diag_reslice_info", "tnco_hip_diag_fw_stats", "tnco_hip_sync", "tnco_hip_get_costs", "tnco_hip_get_tree",
    "tnco_hip_get_caches", "tnco_hip_validate", "tnco_hip_get_prng", "tnco_hip_set_prng", "tnco_hip_get_prng_many", "tnco_hip_set_prng_many",
    "tnco_hip_best", "tnco_hip_min_cost_device", "tnco_hip_get_trees", "tnco_hip_linear_paths", "tnco_hip_linear_paths_ssa", "tnco_hip_diag_counters", "tnco_hip_diag_moves", "tnco_hip_diag_full_copies",
    "tnco_hip_diag_kernel_time", "tnco_hip_diag_kernel_times", "tnco_hip_diag_stage_cycles",
    "tnco_hip_diag_launch_groups", "tnco_hip_diag_device_bytes", "tnco_hip_set_stream", "tnco_hip_destroy", "tnco_hip_release_cached", "tnco_hip_diag_cached_bytes", "tnco_hip_random_trees", "tnco_hip_greedy_trees",
    "tnco_hip_greedy_trees_device", "tnco_hip_diag_greedy_device_supported", "tnco_hip_diag_greedy_device_redone", "tnco_hip_greedy_device_release", "tnco_hip_copy_to_host", "tnco_hip_diag_greedy_cost_key",
    "tnco_hip_comm_unique_id", "tnco_hip_comm_init", "tnco_hip_comm_destroy", "tnco_hip_comm_allreduce_min", "tnco_hip_comm_allgather",
    "tnco_hip_comm_barrier", "tnco_hip_comm_last_error",
    "tnco_hip_device_name", "tnco_hip_device_count", "tnco_hip_last_error", "tnco_hip_version",
]

_lib = None


def torch_first() -> None:
    """Initialise PyTorch's bundled HIP runtime (see load()); a no-op without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception as e:  # noqa: BLE001 -- reported, not swallowed: the ordering problem would come back silently
        warnings.warn(f"tnco_amd: could not initialise torch's HIP runtime before libtnco_hip.so ({e!r}); "
                      "torch may report 'No HIP GPUs are available' later in this process.")


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not _PATH.exists():
        raise ImportError(
            f"{_PATH} is missing: build it with `make -C tnco_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "tnco_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP / HSA runtime (torch/lib/libamdhip64.so, no soname) next to the
    # system one this library links (/opt/rocm/lib/libamdhip64.so.7): both end up in a process that
    # uses torch.  They coexist when torch's runtime initialises FIRST; the other way round torch
    # reports "No HIP GPUs are available".  So a process that HAS imported torch gets torch's runtime
    # initialised here, before the library touches the GPU -- and only such a process: a rank of a
    # torchrun launch that talks through the native communicator (parallel.NativeComm) never imports
    # torch and holds ONE HIP runtime; the torch.distributed transport imports torch before it loads
    # the library (parallel._dist).
    # A process told to talk through torch.distributed (TNCO_COMM=torch*) WILL import torch later: its runtime goes
    # first even if the caller loaded this library before anything imported torch.
    will_use_torch = os.environ.get("TNCO_COMM", "").startswith("torch") and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if not os.environ.get("TNCO_HIP_NO_TORCH_FIRST") and ("torch" in sys.modules or will_use_torch):
        torch_first()
    L = C.CDLL(str(_PATH))
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    L.tnco_hip_create.argtypes = [C.POINTER(Desc), C.POINTER(vp)]
    L.tnco_hip_run.argtypes = [vp, C.c_int, vp, i64]
    L.tnco_hip_run_fw.argtypes = [vp, C.c_int, vp, i64, i64, i64]
    L.tnco_hip_get_slices.argtypes = [vp, i64, vp, vp]
    L.tnco_hip_get_slices_many.argtypes = [vp, i64, vp, vp, vp]
    L.tnco_hip_diag_reslice_info.argtypes = [vp, vp, vp]
    L.tnco_hip_diag_fw_stats.argtypes = [vp, vp]
    L.tnco_hip_sync.argtypes = [vp]
    L.tnco_hip_get_costs.argtypes = [vp, vp, vp]
    L.tnco_hip_get_tree.argtypes = [vp, i64, C.c_int, vp, vp, vp, vp]
    L.tnco_hip_get_caches.argtypes = [vp, i64, vp, vp, vp]
    L.tnco_hip_validate.argtypes = [vp, dbl, C.POINTER(i64), C.POINTER(i64)]
    L.tnco_hip_get_prng.argtypes = [vp, i64, vp]
    L.tnco_hip_set_prng.argtypes = [vp, i64, vp]
    L.tnco_hip_get_prng_many.argtypes = [vp, i64, vp, vp]
    L.tnco_hip_set_prng_many.argtypes = [vp, i64, vp, vp]
    L.tnco_hip_best.argtypes = [vp, i64, vp, vp]
    L.tnco_hip_min_cost_device.argtypes = [vp, vp]
    L.tnco_hip_get_trees.argtypes = [vp, i64, vp, C.c_int, vp, vp]
    L.tnco_hip_linear_paths.argtypes = [i32, i32, vp, i64, vp, vp, i32]
    L.tnco_hip_linear_paths_ssa.argtypes = [i32, i32, i64, vp, vp, i32]
    L.tnco_hip_diag_counters.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 4
    L.tnco_hip_diag_moves.argtypes = [vp, vp]
    L.tnco_hip_diag_stage_cycles.argtypes = [vp, vp]
    L.tnco_hip_diag_full_copies.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.tnco_hip_diag_kernel_time.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.c_int]
    L.tnco_hip_diag_kernel_times.argtypes = [vp, vp, vp, C.c_int]
    L.tnco_hip_diag_launch_groups.argtypes = [vp]
    L.tnco_hip_diag_device_bytes.argtypes = [vp]
    L.tnco_hip_diag_device_bytes.restype = i64
    L.tnco_hip_set_stream.argtypes = [vp, vp]
    L.tnco_hip_destroy.argtypes = [vp]
    L.tnco_hip_destroy.restype = None
    L.tnco_hip_release_cached.argtypes = []
    L.tnco_hip_release_cached.restype = None
    L.tnco_hip_diag_cached_bytes.argtypes = []
    L.tnco_hip_diag_cached_bytes.restype = C.c_uint64
    L.tnco_hip_random_trees.argtypes = [i32, i32, vp, vp, i64, vp, vp, i32]
    L.tnco_hip_greedy_trees.argtypes = [i32, i32, vp, vp, vp, i64, vp, vp, vp, i32]
    L.tnco_hip_greedy_trees_device.argtypes = [i32, i32, i32, vp, vp, vp, i64, vp, vp, vp, vp, i32]
    L.tnco_hip_copy_to_host.argtypes = [vp, vp, C.c_uint64]
    L.tnco_hip_diag_greedy_cost_key.argtypes = [i32, i32, i32]
    L.tnco_hip_diag_greedy_cost_key.restype = C.c_uint64
    L.tnco_hip_diag_greedy_device_supported.argtypes = [i32, i32, vp]
    L.tnco_hip_greedy_device_release.argtypes = []
    L.tnco_hip_greedy_device_release.restype = None
    L.tnco_hip_diag_greedy_device_redone.argtypes = []
    L.tnco_hip_diag_greedy_device_redone.restype = i64
    L.tnco_hip_comm_unique_id.argtypes = [vp]
    L.tnco_hip_comm_init.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.tnco_hip_comm_destroy.argtypes = [vp]
    L.tnco_hip_comm_destroy.restype = None
    L.tnco_hip_comm_allreduce_min.argtypes = [vp, vp, dbl, C.POINTER(dbl)]
    L.tnco_hip_comm_allgather.argtypes = [vp, vp, vp, C.c_uint64]
    L.tnco_hip_comm_barrier.argtypes = [vp]
    L.tnco_hip_comm_last_error.restype = C.c_char_p
    L.tnco_hip_device_name.argtypes = [C.c_int, vp, C.c_int]
    L.tnco_hip_device_count.restype = C.c_int
    L.tnco_hip_last_error.restype = C.c_char_p
    L.tnco_hip_version.restype = C.c_char_p
    _lib = L
    return L


def check(rc: int) -> None:
    """Map a status code to the exception class the reference would raise."""
    if rc == OK:
        return
    msg = load().tnco_hip_last_error().decode()
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENOTIMPL:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
