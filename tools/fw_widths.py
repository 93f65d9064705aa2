"""Finite-width throughput against max_width, and what the re-slices did (tnco_hip_diag_fw_stats).

    python tools/fw_widths.py [--layout supremacy] [--widths 28,32,40] [--replicas 32768] [--sweeps 500] [--chunk 100]

Per chunk of sweeps: move evaluations / s, replica re-slices in the re-priced (one-wavefront) form, the
share of them left to the full rebuild and why, indices changed by the last proposal (median / max), sliced
indices (median / max), best log10(cost).  --im S also runs S sweeps of the infinite-memory optimizer on the
same network (best log10(flops) without a width limit).  Initial trees as the reference draws them.
"""
import argparse
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import core, synthetic  # noqa: E402


def popcounts(a):
    return np.unpackbits(np.ascontiguousarray(a).view(np.uint8), axis=-1).sum(axis=-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="supremacy")
    ap.add_argument("--network", default="sycamore", help="sycamore | cz:<depth>:<fuse|raw> | regular:<n> | hyper:<n>:<inds>")
    ap.add_argument("--frac", default=None, help="max_width as fractions of the median width of the greedy starts, e.g. 0.7,0.85 (instead of --widths)")
    ap.add_argument("--widths", default="28,32,40")
    ap.add_argument("--replicas", type=int, default=32768)
    ap.add_argument("--sweeps", type=int, default=500)
    ap.add_argument("--chunk", type=int, default=100)
    ap.add_argument("--update-slices", type=int, default=10)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--im", type=int, default=0)
    a = ap.parse_args()
    if a.network == "sycamore":
        p = synthetic.sycamore_problem(a.depth, a.layout)
        name = f"Sycamore-53 depth-{a.depth} ({a.layout})"
    elif a.network.startswith("cz:"):
        _, depth, fz = a.network.split(":")
        ts, dims, out = synthetic.sycamore53_cz_tn(int(depth), None if fz == "raw" else float(fz))
        p = synthetic.Problem(ts, 2, out)
        name = f"Sycamore-53 lattice, CZ gates decomposed into hyper-indices, depth {depth}, fuse {fz}"
    elif a.network.startswith("regular:"):
        p = synthetic.regular_problem(int(a.network.split(":")[1]), 11)
        name = a.network
    else:
        _, n, ni = a.network.split(":")
        ts, dims, out = synthetic.random_hyper_tn(int(n), int(ni), k=3, seed=5)
        p = synthetic.Problem(ts, 2, out)
        name = a.network
    seeds = synthetic.replica_seeds(a.replicas)
    print(f"# {name}: {p.n} tensors, {p.n_inds} indices, {p.W} mask words, {a.replicas} replicas", flush=True)
    t0 = time.perf_counter()
    links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, device=0)
    print(f"# greedy initial trees: {time.perf_counter() - t0:.2f} s", flush=True)
    widths8 = []
    for r in range(8):
        m = p.node_masks(links[r, 0], links[r, 1])
        widths8.append(int(popcounts(m).max()))
    print(f"# widths of the first 8 greedy trees: {sorted(widths8)}; indices held by > 2 tensors: "
          f"{sum(len(h) > 2 for h in p.holders)} of {p.n_inds}", flush=True)
    if a.frac:
        a.widths = ",".join(str(int(round(float(f) * float(np.median(widths8))))) for f in a.frac.split(","))
    betas = synthetic.linear_betas(0.0, 100.0, a.sweeps)
    if a.im:
        with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, device=0) as opt:
            b = synthetic.linear_betas(0.0, 100.0, a.im)
            opt.sync()
            t0 = time.perf_counter()
            opt.run(b)
            opt.sync()
            dt = time.perf_counter() - t0
            c = opt.counters()
            tot, mn = opt.costs()
            print(f"infinite memory: {a.im} sweeps {dt:.2f} s, {c['moves'] / dt:.3e} move-evals/s, "
                  f"best log10(flops) {np.log10(mn.min()):.2f}, median {np.log10(np.median(mn)):.2f}", flush=True)
    for mw in [float(x) for x in a.widths.split(",")]:
        t0 = time.perf_counter()
        opt = core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, device=0, max_width=mw)
        opt.sync()
        print(f"## max_width {mw:g}: create (incl. initial slicing) {time.perf_counter() - t0:.2f} s", flush=True)
        prev_c, prev_s = opt.counters(), opt.fw_stats()
        for s0 in range(0, a.sweeps, a.chunk):
            t0 = time.perf_counter()
            opt.run(betas[s0:s0 + a.chunk], update_slices_every=a.update_slices)
            opt.sync()
            dt = time.perf_counter() - t0
            c, s = opt.counters(), opt.fw_stats()
            d = {k: s[k] - prev_s[k] for k in s}
            mv = c["moves"] - prev_c["moves"]
            acc = (c["accepted"] - prev_c["accepted"]) / max(mv, 1)
            prev_c, prev_s = c, s
            try:
                how, nch = opt.reslice_info()
                nchv = nch[nch >= 0]
                chg = f"changed med {np.median(nchv) if len(nchv) else -1:.0f} max {nchv.max() if len(nchv) else -1}"
            except ValueError:
                chg = "changed n/a"
            sl = popcounts(opt.slices_many(np.arange(min(a.replicas, 4096)))[0])
            tot, mn = opt.costs()
            rp = max(d["repriced"], 1)
            print(f"sweeps {s0:5d}+{a.chunk}: {mv / dt:.3e} moves/s  accept {acc:.2f}  repriced {d['repriced']:9d} "
                  f"fell back {d['fell_back'] / rp:.4f} (wide {d['too_many_wide'] / rp:.4f} changed {d['too_many_changed'] / rp:.4f} "
                  f"range {d['cost_range'] / rp:.4f})  full-form {d['full_rebuild_form']:9d}  {chg}  "
                  f"slices med {np.median(sl):.0f} max {sl.max()}  best log10 {np.log10(mn.min()):.2f}", flush=True)
        bad = opt.validate()
        print(f"   validate: {bad}", flush=True)
        opt.close()


if __name__ == "__main__":
    main()
