"""Throughput of the finite-width kernels on the BASELINE config-5 network (the Sycamore-53 supremacy circuit at
depth 20: 536 tensors, 913 indices), memory-constrained, from random-Kruskal starts.

    python tools/time_fw.py [--replicas 4096] [--sweeps 100] [--max-width 40] [--cpu-sample 32]

--cpu-sample N also runs N replicas of the same problem through the CPU oracle (oracle/, the plain-C
restatement of the reference's finite-width optimizer; a measurement tool, like bench.py's
cpu_baseline leg), one replica per host thread, and checks that they end at the GPU's costs.
"""
import argparse
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import core, ctree, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=4096)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--max-width", type=float, default=40)
    ap.add_argument("--update-slices", type=int, default=10)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=0)
    a = ap.parse_args()
    ts, dims, out = synthetic.sycamore53_tn(a.depth)
    n_inds = 1 + max(i for xs in ts for i in xs)
    lm = ctree.pack_masks(ts, n_inds)
    seeds = np.arange(1, a.replicas + 1, dtype=np.uint32)
    t0 = time.perf_counter()
    links = core.random_trees(ts, n_inds, seeds)
    t1 = time.perf_counter()
    opt = core.BatchedOptimizer(lm, links, seeds, n_inds=n_inds, max_width=a.max_width)
    opt.sync()
    t2 = time.perf_counter()
    betas = np.linspace(0, 100, a.sweeps)
    opt.run(betas, update_slices_every=a.update_slices)
    opt.sync()
    t3 = time.perf_counter()
    c = opt.counters()
    tot, mn = opt.costs()
    print(f"{len(ts)} tensors, {n_inds} indices, {a.replicas} replicas, max_width {a.max_width}")
    print(f"initial trees {t1 - t0:.2f} s, create (incl. initial slicing) {t2 - t1:.2f} s, "
          f"{a.sweeps} sweeps {t3 - t2:.2f} s")
    print(f"moves {c['moves']:.3e}  accepted {c['accepted'] / max(c['moves'], 1):.3f}  "
          f"-> {c['moves'] / (t3 - t2):.3e} move-evals/s;  best log2(cost) {np.log2(mn.min()):.2f}, "
          f"slices of the best: {int(np.unpackbits(opt.slices(int(mn.argmin()))[1].view(np.uint8)).sum())}")
    if a.cpu_sample > 0:
        cpu_leg(a, ts, n_inds, lm, links, seeds, betas, tot)


def cpu_leg(a, ts, n_inds, lm, links, seeds, betas, gpu_total):
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.build()
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
    from bench import usable_cores  # (affinity mask capped by the cgroup quota)
    k = min(a.cpu_sample, a.replicas)
    cores = usable_cores()

    def make(r):
        l, rr, p = (np.ascontiguousarray(links[r, j]) for j in range(3))
        inds = ctree.derive_inds(l, rr, lm, None)
        return orc.Oracle(l, rr, p, inds, n_inds=n_inds, dims=2, seed=int(seeds[r]), max_width=a.max_width)

    def run(o):  # (the C call releases the GIL)
        t = time.perf_counter()
        o.run(orc.PROB_MH, betas, update_slices_every=a.update_slices)
        return o.counters()["moves"], time.perf_counter() - t, o.total_cost

    states = [make(r) for r in range(k)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        out = list(ex.map(run, states))
    wall = time.perf_counter() - t0
    moves = sum(m for m, _, _ in out)
    same = all(c == gpu_total[r] for r, (_, _, c) in enumerate(out))
    print(f"CPU oracle: {k} replicas on {cores} threads, {wall:.2f} s -> {moves / wall:.3e} move-evals/s "
          f"({moves / sum(d for _, d, _ in out):.3e} per core); same final costs as the GPU: {same}")


if __name__ == "__main__":
    main()
