#!/usr/bin/env python3
"""bench.py -- SA move-evaluations/s of the HIP path on synthetic 3-regular tensor networks.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched by torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" = one launch of the sweep kernel: `--sweeps-per-step` calls of Optimizer::update
(include/tnco/optimize/infinite_memory/optimizer.hpp:90-221 of the reference) on EVERY replica
resident on the GPU.  The beta schedule is linear 0 -> 100 over all (W + K) * sweeps_per_step
sweeps, as tnco/app/infinite_memory/sa.py:147-156 builds it.  Inputs (trees, masks, PRNG state) are
resident in HBM before the timed region.  Workload at N = 1: BASELINE.json configs[2], the
configuration the metric is quoted on: 512-leaf random 3-regular TN (bond dim 2), 65536 replicas.
For N > 1 every rank owns 65536 replicas of the same TN (weak scaling, configs[3] at N = 8);
the only collective is one RCCL all-reduce(min) of the best cost inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_move(W: int, a: float, q: float) -> float:
    """B_move = 56W + 80 + a(24W + 64) + 8(2 + q)."""
    return 56 * W + 80 + a * (24 * W + 64) + 8 * (2 + q)


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    c = min(c, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    c = min(c, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return c


def cpu_baseline(prob, links, seeds, betas, n_sample, cores):
    """The oracle (plain-C port of the reference algorithm) on the host cores, bounded sample.

    Only the update loops are timed (OpenMP over replicas inside oracle/tnco_oracle.c); tree
    flattening and cache construction are setup, as on the GPU side."""
    from oracle import oracle as orc
    orc.build()
    dt, _tot, mn, mv = orc.run_batch(links[:n_sample], prob.leaf_masks, seeds[:n_sample], betas,
                                     n_inds=prob.n_inds, dims=2, n_threads=cores)
    moves = int(mv.sum())
    return moves / dt, moves, dt, mn


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sweeps-per-step", type=int, default=100)
    ap.add_argument("--leaves", type=int, default=512)
    ap.add_argument("--replicas", type=int, default=65536, help="replicas per GPU")
    ap.add_argument("--graph-seed", type=int, default=11)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="replicas timed on the CPU oracle (0 = skip, -1 = as many as take ~15 s)")
    ap.add_argument("--validate", action="store_true", help="device-side is_valid() of every replica after the run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from tnco_amd import core, parallel
    from tests import helpers as H

    n, R = args.leaves, args.replicas
    prob = H.regular_problem(n, graph_seed=args.graph_seed)
    all_seeds = H.replica_seeds(R * world, S=0)
    seeds = all_seeds[rank * R:(rank + 1) * R]
    links = core.random_trees(prob.ts_inds, prob.n_inds, seeds)
    total_sweeps = (args.warmup + args.steps) * args.sweeps_per_step
    betas = H.linear_betas(0.0, 100.0, total_sweeps)

    opt = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, dims=2, device=local_rank)
    sps = args.sweeps_per_step

    def barrier():
        opt.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for s in range(args.warmup):
        opt.run(betas[s * sps:(s + 1) * sps])
    barrier()
    c0 = opt.counters()
    opt.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        opt.run(betas[s * sps:(s + 1) * sps])
    best = parallel.global_best(opt, rank=rank, world=world, device=local_rank)
    barrier()
    dt = time.perf_counter() - t0
    c1 = opt.counters()
    kernel_ms, launches = opt.kernel_time_ms()

    moves = c1["moves"] - c0["moves"]
    acc = c1["accepted"] - c0["accepted"]
    rp = c1["random_picks"] - c0["random_picks"]
    stats = torch.tensor([dt, float(moves), float(acc), float(rp), kernel_ms], dtype=torch.float64)
    if world > 1:
        stats = stats.cuda()
        tmax = stats[[0, 4]].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:4].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt, kernel_ms = float(tmax[0]), float(tmax[1])
        moves, acc, rp = (float(x) for x in tot)
    n_bad = None
    if args.validate:
        n_bad = opt.validate()[0]

    if rank == 0:
        a = acc / max(moves, 1)
        q = rp / max(moves, 1)
        bmove = algorithmic_bytes_per_move(prob.W, a, q)
        moves_per_launch_gpu = moves / world / max(launches, 1)
        avg_launch_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = bmove * moves_per_launch_gpu / avg_launch_s / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "SA move-evaluations/s (whole node) + best log10(flops) vs ref, 512-leaf TN",
            "value": moves / dt,
            "unit": "move-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{n}-leaf random 3-regular TN (bond dim 2, {prob.n_inds} indices, "
                            f"{prob.W} mask words), {R} replicas per GPU, "
                            f"{sps} SA sweeps per step, beta linear 0->100 over {total_sweeps} sweeps, "
                            "Metropolis-Hastings, float64 cost",
                "replicas_total": R * world,
                "sweeps_per_step": sps,
                "moves_timed": moves,
                "accept_rate": a,
                "random_pick_rate": q,
                "best_log10_flops": float(np.log10(best)),
                "improvements_timed": c1["improved"] - c0["improved"],
                "full_tree_copies_timed": c1["full_copies"] - c0["full_copies"],
                "validated_bad_replicas": n_bad,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "sa_run_kernel",
                "algorithmic_bytes_per_move": bmove,
                "avg_launch_ms": avg_launch_s * 1e3,
                "launches": launches,
            },
        }
        if args.cpu_sample != 0 and world == 1:
            cores = usable_cores()
            if args.cpu_sample > 0:
                ns = min(args.cpu_sample, R)
            else:  # auto: a probe sets the sample so that the timed run is ~15 s of CPU work
                probe = min(256, R)
                pv, pm, _pt, _ = cpu_baseline(prob, links, seeds, betas, probe, cores)
                ns = int(min(R, max(probe, 15.0 * pv / (pm / probe))))
            v, m, t, cpu_min = cpu_baseline(prob, links, seeds, betas, ns, cores)
            gpu_min = opt.costs()[1][:ns]
            out["config"]["cpu_sample_min_cost_bit_exact"] = bool(np.array_equal(cpu_min, gpu_min))
            out["cpu_baseline"] = {
                "value": v, "unit": "move-evals/s", "cores": cores, "kind": "port",
                "sample": f"oracle/tnco_oracle.c (plain-C restatement), {ns} of the same replicas, full "
                          f"{total_sweeps}-sweep schedule each, {cores} threads; {m} moves in {t:.1f} s",
            }
        print(json.dumps(out), flush=True)
    opt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
