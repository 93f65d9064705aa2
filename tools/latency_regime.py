"""Where the drop-in stops paying: wall time of ONE optimize() call against the number of runs.

The reference's own usage is n_runs = 8 ... 1000 (README.md:93-106 runs 8); a replica of the GPU path advances one
dependent memory round trip at a time, a host core runs it several times faster -- so below some number of runs the
CPU path wins.  For n_runs in {8, 64, 512, 4096} x {64, 512} leaves x `--sweeps` sweeps (10 000 by default):
  GPU:  app.Optimizer(method='sa', seed=0).optimize(<3-regular network>, betas=(0, 100), n_steps, n_runs, fuse=None),
        the whole call (initial trees, create, sweeps, best paths back as Python objects); and the sweeps alone through
        the C ABI (tnco_hip_run ... sync);
  CPU:  the oracle (oracle/tnco_oracle.c, the plain-C restatement of the reference's update loop; `kind: port`) over
        the same runs on all host threads, update loops only (initial trees and result assembly not counted: the CPU
        side is favoured).
Prints one table row per case and the break-even n_runs per network size (log-linear interpolation of the two walls).
Run on the GPU box: python tools/latency_regime.py > gpurun_out/r05/latency_regime.txt
"""
import argparse
import math
import pathlib
import sys
import time
import warnings

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tnco_amd import core, synthetic as syn  # noqa: E402
from tnco_amd.app import Optimizer  # noqa: E402

warnings.simplefilter("ignore")


def spec_of(n, seed):
    ts, _d, _ = syn.random_regular_tn(n, 3, seed)
    n_inds = max(max(x) for x in ts) + 1
    return [(2, *[f"t{t}" for t in range(n) if k in ts[t]]) for k in range(n_inds)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=10000)
    ap.add_argument("--runs", default="8,64,512,4096")
    ap.add_argument("--leaves", default="64,512")
    ap.add_argument("--cpu-budget", type=float, default=40.0, help="skip the CPU leg of a case expected to take longer (seconds)")
    a = ap.parse_args()
    from bench import usable_cores
    from oracle import oracle as orc
    orc.build()
    cores = usable_cores()
    print(f"# {a.sweeps} sweeps per run, beta 0 -> 100, Metropolis, float64; CPU = oracle (port) on {cores} threads, update loops only")
    print("| leaves | n_runs | GPU optimize() wall s | GPU sweeps only s | GPU move-evals/s (sweeps) | CPU s | CPU move-evals/s | GPU optimize() / CPU |")
    print("|---|---|---|---|---|---|---|---|")
    Optimizer(method="sa", seed=0).optimize(spec_of(64, 7), betas=(0, 100), n_steps=10, n_runs=8, fuse=None)  # warm-up
    for n in [int(x) for x in a.leaves.split(",")]:
        gseed = 7 if n == 64 else 11
        prob = syn.regular_problem(n, graph_seed=gseed)
        spec = spec_of(n, gseed)
        betas = syn.linear_betas(0.0, 100.0, a.sweeps)
        rows, cpu_rate = [], None
        for R in [int(x) for x in a.runs.split(",")]:
            t0 = time.perf_counter()
            Optimizer(method="sa", seed=0).optimize(spec, betas=(0, 100), n_steps=a.sweeps, n_runs=R, fuse=None, top_k=min(R, 16))
            wall = time.perf_counter() - t0
            seeds = syn.replica_seeds(R)
            links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
            with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds) as gpu:
                gpu.run(betas[:10]); gpu.sync()
                m0 = gpu.counters()["moves"]
                t0 = time.perf_counter()
                for s in range(10, a.sweeps, 1000):
                    gpu.run(betas[s:s + 1000])
                gpu.sync()
                gs = time.perf_counter() - t0
                gmoves = gpu.counters()["moves"] - m0
                gmin = gpu.costs()[1]
            est = None if cpu_rate is None else gmoves / cpu_rate
            if est is not None and est > a.cpu_budget:
                cs, crate, ratio = None, cpu_rate, wall / est
                rows.append((R, wall, est))
                print(f"| {n} | {R} | {wall:.3f} | {gs:.3f} | {gmoves / gs:.3e} | ~{est:.1f} (extrapolated) | {crate:.3e} | {ratio:.2f} |", flush=True)
                continue
            dt, _tot, mn, mv = orc.run_batch(links, prob.leaf_masks, seeds, betas, n_inds=prob.n_inds, dims=2, n_threads=cores)
            assert np.array_equal(mn, gmin), "GPU and oracle disagree"
            cpu_rate = float(mv.sum()) / dt if R >= cores else cpu_rate or float(mv.sum()) / dt
            rows.append((R, wall, dt))
            print(f"| {n} | {R} | {wall:.3f} | {gs:.3f} | {gmoves / gs:.3e} | {dt:.3f} | {float(mv.sum()) / dt:.3e} | {wall / dt:.2f} |", flush=True)
        be = None
        for (r0, g0, c0), (r1, g1, c1) in zip(rows, rows[1:]):
            if (g0 - c0) > 0 >= (g1 - c1):  # the GPU call is slower at r0, not slower at r1
                f0, f1 = math.log(g0 / c0), math.log(g1 / c1)
                be = math.exp(math.log(r0) + (math.log(r1) - math.log(r0)) * f0 / (f0 - f1))
        if be is None:
            be = f"< {rows[0][0]}" if rows[0][1] <= rows[0][2] else f"> {rows[-1][0]}"
        else:
            be = f"~{be:.0f}"
        print(f"# break-even at {n} leaves, {a.sweeps} sweeps: n_runs {be} (optimize() wall == {cores}-thread oracle)", flush=True)


if __name__ == "__main__":
    main()
