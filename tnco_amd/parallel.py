"""Replica sharding over the GPUs of one node and the best-cost reduction.

Replaces tnco/parallel.py (joblib/loky process fan-out + SharedMemory buffers,
/root/reference/tnco/parallel.py:111-368): replicas never interact, so rank k of
`world` owns a contiguous block of the run list and the only exchange is the
reduction of the best cost (RCCL all-reduce(min) over xGMI when the process
group's backend is "nccl"; gloo in the CPU tests) plus a broadcast of the
winning tree when a caller wants it.
"""
from __future__ import annotations

import numpy as np

__all__ = ["shard_bounds", "global_best", "global_winner"]


def shard_bounds(n_runs: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of the run list owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("'rank' / 'world' are not valid.")
    base, extra = divmod(int(n_runs), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def rank_world() -> tuple[int, int]:
    """(rank, world) of the default process group, (0, 1) when torch.distributed is not in use."""
    try:
        _torch, dist = _dist()
    except ImportError:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def local_device() -> int:
    """GPU ordinal of this rank: LOCAL_RANK under torchrun, else 0."""
    import os
    return int(os.environ.get("LOCAL_RANK", "0"))


def merge_heads(local: list, k: int, rank: int, world: int) -> list:
    """Every rank contributes its best runs as tuples sorted by (cost, global run id, ...); all ranks
    receive the k best overall in that order (the head of `sorted(results)`,
    tnco/app/infinite_memory/sa.py:257)."""
    if world == 1:
        return sorted(local, key=lambda t: (t[0], t[1]))[:k]
    _torch, dist = _dist()
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    merged = [t for part in gathered for t in part]
    return sorted(merged, key=lambda t: (t[0], t[1]))[:k]


def _tensor_device(dist, device):
    import torch
    if dist.get_backend() == "nccl":
        return torch.device("cuda", device)
    return torch.device("cpu")


def global_best(opt_or_cost, rank: int = 0, world: int = 1, device: int = 0, grouped: bool | None = None) -> float:
    """min over all ranks of the local best min_total_cost.

    Given an optimizer handle and an RCCL group, the local minimum is reduced on the device
    (tnco_hip_min_cost_device) straight into the tensor the all-reduce(min) runs on: 8 bytes over
    xGMI, no host round trip before the collective.  `grouped` (default: world > 1) = go through the
    process group; a group of ONE rank takes the same code path (tests/test_gpu_two_ranks.py runs
    RCCL that way on a 1-GPU box)."""
    is_opt = hasattr(opt_or_cost, "best")
    if not (world > 1 if grouped is None else grouped):
        return float(opt_or_cost.best(1)[0][0]) if is_opt else float(opt_or_cost)
    torch, dist = _dist()
    dev = _tensor_device(dist, device)
    if is_opt and dev.type == "cuda" and hasattr(opt_or_cost, "min_cost_to_device"):
        t = torch.empty(1, dtype=torch.float64, device=dev)
        opt_or_cost.min_cost_to_device(t.data_ptr())
    else:
        c = float(opt_or_cost.best(1)[0][0]) if is_opt else float(opt_or_cost)
        t = torch.tensor([c], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t[0])


def global_winner(cost: float, global_id: int, payload: np.ndarray | None, rank: int, world: int,
                  device: int = 0):
    """(best cost, its global run id, winner's payload) on every rank.

    Ties go to the lowest global id (the order `sorted(results)` keeps,
    tnco/app/infinite_memory/sa.py:257).  `payload` is an int32 array of equal
    shape on every rank (e.g. the best tree's links), broadcast from the winner.
    """
    if world == 1:
        return cost, global_id, payload
    torch, dist = _dist()
    dev = _tensor_device(dist, device)
    mine = torch.tensor([cost, float(global_id)], dtype=torch.float64, device=dev)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    pairs = [(float(v[0]), int(v[1]), k) for k, v in enumerate(allv)]
    best_cost, best_id, src = min(pairs)
    if payload is not None:
        t = torch.from_numpy(np.ascontiguousarray(payload, np.int32)).to(dev)
        dist.broadcast(t, src=src)
        payload = t.cpu().numpy()
    return best_cost, best_id, payload
