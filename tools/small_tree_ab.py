"""A/B of the LDS-resident small-tree kernel (csrc/sa_small.h) against the HBM kernel, sweeps only.

For 3-regular networks of 64 and 84 leaves (<= 2 mask words: sa_small_kernel) -- or `--leaves 128,256,512`: sa_lds_kernel --
and a range of replica counts: the same seeded runs once through
the library in the tree (which picks the LDS kernel while one round of its blocks holds every replica) and once through
build_variants/lib_nosmall.so and lib_allsmall.so (`make -C tnco_amd/csrc nosmall allsmall`: the same sources with
-DTNCO_NO_SMALL_TREE, always the HBM kernel, and with -DTNCO_SMALL_TREE_ALWAYS, the LDS kernel whatever the number of
replicas) -- move-evals/s of the sweeps, which kernel the tree's library chose, and whether the two end states are the same
bit for bit (counters, current and minimum costs of every replica, the best tree).
Run on the GPU box: python tools/small_tree_ab.py > gpurun_out/r05/small_tree_ab.txt
"""
import argparse
import hashlib
import json
import os
import pathlib
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def child(a):
    import numpy as np
    from tnco_amd import core, synthetic as syn
    for n in [int(x) for x in a.leaves.split(",")]:
        if a.hyper:  # a network with hyper-indices: n tensors of 3 indices each out of 1.5 n, 5 of them open
            prob = syn.Problem(*(lambda ts, d, out: (ts, 2, out))(*syn.random_hyper_tn(n, 3 * n // 2, k=3, n_output=5, seed=n)))
        else:
            prob = syn.regular_problem(n, graph_seed=7 if n == 64 else 11)
        betas = syn.linear_betas(0.0, 100.0, a.sweeps)
        for R in [int(x) for x in a.runs.split(",")]:
            seeds = syn.replica_seeds(R)
            links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds)
            with core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds) as gpu:
                gpu.run(betas[:10]); gpu.sync()
                m0 = gpu.counters()["moves"]
                t0 = time.perf_counter()
                for s in range(10, len(betas), 1000):
                    gpu.run(betas[s:s + 1000])
                gpu.sync()
                dt = time.perf_counter() - t0
                c = gpu.counters()
                cur, mn = gpu.costs()
                hs = hashlib.sha256()
                for x in (np.asarray(cur), np.asarray(mn), *gpu.tree(int(np.argmin(mn)), which_min=True)):
                    hs.update(np.ascontiguousarray(x).tobytes())
                hs.update(repr((c["moves"], c["accepted"], c["improved"], c["random_picks"])).encode())
                print(json.dumps(dict(n=n, W=int(prob.leaf_masks.shape[1]), R=R, rate=(c["moves"] - m0) / dt, lds=gpu.launch_groups == 0,
                                      state=hs.hexdigest(), bad=int(gpu.validate()[0]))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=2010)
    ap.add_argument("--runs", default="64,512,2048,4096,8192,16384,32768,65536")
    ap.add_argument("--leaves", default="64,84")
    ap.add_argument("--hyper", action="store_true", help="networks with hyper-indices (random_hyper_tn) instead of 3-regular ones")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    variant = ROOT / "build_variants" / "lib_nosmall.so"
    always = ROOT / "build_variants" / "lib_allsmall.so"
    assert variant.exists() and always.exists(), "make -C tnco_amd/csrc nosmall allsmall"
    res = {}
    for name, lib in (("tree", None), ("hbm", str(variant)), ("lds", str(always))):
        env = dict(os.environ)
        env.pop("TNCO_HIP_LIB", None)
        if lib:
            env["TNCO_HIP_LIB"] = lib
        out = subprocess.run([sys.executable, __file__, "--child", "--sweeps", str(a.sweeps), "--runs", a.runs, "--leaves", a.leaves] + (["--hyper"] if a.hyper else []),
                             env=env, capture_output=True, text=True, stdin=subprocess.DEVNULL)
        if out.returncode:
            sys.exit(out.stderr[-3000:])
        res[name] = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    print(f"# {a.sweeps} sweeps, beta 0 -> 100, Metropolis, float64, {'networks with hyper-indices (n tensors x 3 indices out of 1.5 n)' if a.hyper else '3-regular networks'}; 'library' = tnco_amd/libtnco_hip.so, 'HBM' / 'LDS' = the same sources with -DTNCO_NO_SMALL_TREE / -DTNCO_SMALL_TREE_ALWAYS")
    print("| leaves | mask words | replicas | HBM kernel move-evals/s | LDS kernel move-evals/s | library move-evals/s | library's kernel | library / HBM | same end states | replicas failing validate |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for t, h, l in zip(res["tree"], res["hbm"], res["lds"]):
        assert (t["n"], t["R"]) == (h["n"], h["R"]) == (l["n"], l["R"]) and not h["lds"]
        lds_rate = f"{l['rate']:.3e}" if l["lds"] else "(a tree does not fit)"
        print(f"| {t['n']} | {t['W']} | {t['R']} | {h['rate']:.3e} | {lds_rate} | {t['rate']:.3e} | {'LDS-resident' if t['lds'] else 'HBM'} | {t['rate'] / h['rate']:.2f} | "
              f"{t['state'] == h['state'] == l['state']} | {t['bad'] + h['bad'] + l['bad']} |")


if __name__ == "__main__":
    main()
