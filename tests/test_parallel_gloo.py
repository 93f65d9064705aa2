"""world_size-2 CPU test (gloo) of the multi-GPU path's host logic: shard, reduce, merge, broadcast."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tnco_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.rank_world() == (rank, world)
        n_runs = 11
        lo, hi = parallel.shard_bounds(n_runs, world, rank)
        costs = np.array([50.0, 7.0, 9.0, 30.0, 7.0, 12.0, 99.0, 8.0, 41.0, 7.5, 60.0])  # run -> cost
        mine = costs[lo:hi]
        best = parallel.global_best(float(mine.min()), rank, world)
        gid = lo + int(np.argmin(mine))
        payload = np.full((3, 5), gid, np.int32)
        wc, wid, wp = parallel.global_winner(float(mine.min()), gid, payload, rank, world)
        local = sorted(((float(c), lo + k, [c], [[(0, 1)]]) for k, c in enumerate(mine)))[:3]
        merged = parallel.merge_heads(local, 3, rank, world)
        q.put((rank, best, wc, wid, wp.tolist(), [(c, g) for c, g, _, _ in merged]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_reduction_and_merge():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, best, wc, wid, wp, merged in out:
        assert best == 7.0 and wc == 7.0 and wid == 1          # tie 7.0 at runs 1 and 4 -> lowest id
        assert wp == [[1] * 5] * 3                              # winner's payload broadcast from rank 0
        assert merged == [(7.0, 1), (7.0, 4), (7.5, 9)]          # same head on every rank


class _StubOptimizer:
    """What bench.py's reduction sees of a handle: best(k) and the work counters."""

    def __init__(self, rank):
        self.rank = rank

    def best(self, k):
        return np.array([10.0 - 3 * self.rank]), np.array([5])


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import bench
    from tnco_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kt = {k: (1.0 + rank if k == "sa_run_kernel" else 0.0, 3 if k == "sa_run_kernel" else 0) for k in bench.KERNELS}
        res = dict(dt=0.5 + 0.25 * rank, moves=1000 * (rank + 1), accepted=100 * (rank + 1), random_picks=10,
                   improved=7, full_copies=0, kt=kt, best=parallel.global_best(_StubOptimizer(rank), rank, world))
        out = bench.reduce_legs(res, world, dist, torch)
        q.put((rank, out["dt"], out["moves"], out["accepted"], out["kt"]["sa_run_kernel"][0], out["best"],
               [(p["rank"], p["moves"]) for p in out["per_rank"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bench_reduction_over_two_ranks():
    """bench.py's N > 1 path without GPUs: the best cost is the min over ranks, the timed region the
    max, the work the sum, and the line lists what every rank did."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, dt, moves, acc, kms, best, per_rank in out:
        assert dt == 0.75 and moves == 3000 and acc == 300 and kms == 2.0 and best == 7.0
        assert per_rank == [(0, 1000.0), (1, 2000.0)]
