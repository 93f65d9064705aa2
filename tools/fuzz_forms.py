"""The two forms of the finite-width re-slice against each other on random mid-size networks (no oracle: sizes the
oracle would take minutes on): one wavefront per replica (TNCO_HIP_FW_WAVE=1; lean, roomier, hyper, 16 / 32 / 64 lanes
per mask, up to 2 048 tensors) == walk + full rebuild (TNCO_HIP_FW_WAVE=0) -- totals, best totals, slices, best slices,
generator states of every replica, and every replica valid on the device.

    python tools/fuzz_forms.py [--cases 24] [--seed 0] [--replicas 1024] [--sweeps 30]
"""
import argparse
import os
import pathlib
import random
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import core, synthetic as syn  # noqa: E402


def network(rng):
    kind = rng.choice(["regular", "regular", "hyper", "cz_fused", "cz_raw"])
    if kind == "regular":
        n = rng.choice([96, 200, 340, 520, 700, 1100, 1500])
        deg = rng.choice([3, 3, 4, 5]) if n <= 700 else 3
        if n * deg % 2:
            n += 1
        ts, d, out = syn.random_regular_tn(n, deg, rng.randrange(1000))
        return f"{deg}-regular {n}", syn.Problem(ts, 2, out)
    if kind == "hyper":
        n = rng.choice([80, 150, 300, 500])
        ts, dims, out = syn.random_hyper_tn(n, int(rng.uniform(1.6, 2.4) * n), k=rng.choice([3, 4, 5]), n_output=rng.choice([0, 3]),
                                            seed=rng.randrange(1000))
        return f"hypergraph {n}", syn.Problem(ts, 2, out)
    if kind == "cz_fused":
        depth = rng.choice([8, 12, 16, 24])
        ts, dims, out = syn.sycamore53_cz_tn(depth, fuse=rng.choice([3, 4, 5]), seed=rng.randrange(100))
        return f"CZ circuit depth {depth} fused", syn.Problem(ts, 2, out)
    depth = rng.choice([4, 8, 12, 20])
    ts, dims, out = syn.sycamore53_cz_tn(depth, fuse=None)
    return f"CZ circuit depth {depth} raw", syn.Problem(ts, 2, out)


def popc(a):
    return np.unpackbits(np.ascontiguousarray(a).view(np.uint8), axis=-1).sum(axis=-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--sweeps", type=int, default=30)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bad = 0
    for case in range(a.cases):
        name, p = network(rng)
        seeds = np.asarray(syn.replica_seeds(a.replicas, S=rng.randrange(10**6)))
        links = core.greedy_trees(p.ts_inds, p.n_inds, seeds, output_mask=p.output_mask, device=0)
        w0 = int(np.median([popc(p.node_masks(links[r, 0], links[r, 1])).max() for r in range(4)]))
        frac = rng.choice([0.5, 0.7, 0.85])
        mw = max(3, int(frac * w0))
        every = rng.choice([3, 5, 10])
        big = rng.choice([None, None, "0", "1"])
        betas = syn.linear_betas(0.0, 60.0, a.sweeps)
        res = []
        for pin in ("1", "0"):
            os.environ["TNCO_HIP_FW_WAVE"] = pin
            if big is None:
                os.environ.pop("TNCO_HIP_FW_BIG", None)
            else:
                os.environ["TNCO_HIP_FW_BIG"] = big
            with core.BatchedOptimizer(p.leaf_masks, links, seeds, n_inds=p.n_inds, output_mask=p.output_mask, max_width=mw) as g:
                for c in range(0, a.sweeps, 10):
                    g.run(betas[c:c + 10], update_slices_every=every)
                v = g.validate()
                st = g.fw_stats()
                res.append((g.costs(), g.slices_many(np.arange(a.replicas)), np.asarray(g.prng_states()), v, st))
        x, y = res
        same = (np.array_equal(x[0][0], y[0][0]) and np.array_equal(x[0][1], y[0][1]) and np.array_equal(x[1][0], y[1][0])
                and np.array_equal(x[1][1], y[1][1]) and np.array_equal(x[2], y[2]) and x[3] == (0, -1) and y[3] == (0, -1))
        st = x[4]
        print(f"{'ok  ' if same else 'FAIL'} {name}: {p.n} tensors, {p.W} words, hyper {any(len(h) > 2 for h in p.holders)}, greedy width {w0}, "
              f"max_width {mw}, every {every}, BIG {big}: re-priced {st['repriced']}, fell back {st['fell_back']} "
              f"(wide {st['too_many_wide']}, changed {st['too_many_changed']}), general form {st['full_rebuild_form']}", flush=True)
        bad += not same
    print(f"{a.cases} cases, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
