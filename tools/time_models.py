"""Throughput of the sweep kernel's other instantiations (diagnostic): hyper-indices, dims that are
not a power of two (cost table), per-index dims (sequential product), float32 cost, sparse legs --
next to the benchmark's fast path (3-regular, dims = 2)."""
import argparse
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import core, ctree, synthetic  # noqa: E402


def run(label, ts, n_inds, R, sweeps, out=(), **kw):
    lm = ctree.pack_masks(ts, n_inds)
    seeds = np.arange(1, R + 1, dtype=np.uint32)
    links = core.random_trees(ts, n_inds, seeds)
    om = ctree.pack_masks([list(out)], n_inds)[0] if len(out) else None
    opt = core.BatchedOptimizer(lm, links, seeds, n_inds=n_inds, output_mask=om, **kw)
    # as bench.py times a leg: warm-up calls, then K calls of `sweeps` sweeps back to back (the library does not wait
    # between calls), one sync at the end -- a single call with a host sync behind it pays the launch tail once per
    # call and reads 25-30 % low
    K = 6
    betas = np.linspace(0, 100, sweeps * (K + 2))
    opt.run(betas[:sweeps])
    opt.run(betas[sweeps:2 * sweeps])
    opt.sync()
    m0 = opt.counters()["moves"]
    t0 = time.perf_counter()
    for k in range(2, K + 2):
        opt.run(betas[k * sweeps:(k + 1) * sweeps])
    opt.sync()
    dt = time.perf_counter() - t0
    print(f"{label:58s} {(opt.counters()['moves'] - m0) / dt:10.3e} move-evals/s", flush=True)
    opt.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=65536)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--only", default=None, help="hyper: the hyper-index network alone (for a rocprofv3 --pmc run)")
    a = ap.parse_args()
    R, S = a.replicas, a.sweeps
    if a.only == "hyper":
        hts, hd, hout = synthetic.random_hyper_tn(512, 768, k=3, n_output=8, seed=3)
        run("hyper-index network, 512 tensors, 768 indices, dims 2", hts, 768, R, S, out=hout)
        return
    ts, _d, _ = synthetic.random_regular_tn(512, 3, 11)
    I = 768
    run("3-regular 512 leaves, dims 2 (benchmark path)", ts, I, R, S)
    ts128, _d, _ = synthetic.random_regular_tn(128, 3, 11)
    run("3-regular 128 leaves, dims 2", ts128, 192, R, S)
    run("  same, float32 cost (512 leaves overflow float32)", ts128, 192, R, S, cost_type="float32")
    run("  same, dims 3 (cost table)", ts, I, R, S, dims=3)
    rng = np.random.RandomState(0)
    run("  same, per-index dims in {2,3,4} (sequential product)", ts, I, R, S, dims=rng.choice([2, 3, 4], size=I).astype(np.uint64))
    run("  same, per-index dims in {2,4,8} (exponent classes)", ts, I, R, S, dims=rng.choice([2, 4, 8], size=I).astype(np.uint64))
    sp = ctree.pack_masks([list(range(0, I, 7))], I)[0]
    run("  same, dims 2, 110 sparse legs, n_projs 1000", ts, I, R, S, sparse_mask=sp, n_projs=1000)
    hts, hd, hout = synthetic.random_hyper_tn(512, 768, k=3, n_output=8, seed=3)
    run("hyper-index network, 512 tensors, 768 indices, dims 2", hts, 768, R, S, out=hout)


if __name__ == "__main__":
    main()
