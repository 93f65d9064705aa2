"""Where do the cycles of one loop iteration of the sweep kernel go?  Needs a library built with
-DTNCO_PROFILE (make -C tnco_amd/csrc profile -> build_variants/lib_profile.so):

    TNCO_HIP_LIB=$PWD/build_variants/lib_profile.so python tools/stage_cycles.py
    tools/build_variant.sh profile4 -DTNCO_PROFILE=4   # cycles inside the greedy pass of the re-slice
    TNCO_HIP_LIB=$PWD/build_variants/lib_profile4.so python tools/stage_cycles.py --fw --fine
(-DTNCO_PROFILE=3: event counts of the re-slice instead of cycles: too-wide tensors scanned, tensors
shuffled and picked from, candidate legs, picks.)
"""
import argparse
import ctypes as C
import sys
import pathlib

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import _lib, core, ctree, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves", type=int, default=512)
    ap.add_argument("--replicas", type=int, default=65536)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--fine", action="store_true")
    ap.add_argument("--fw", action="store_true", help="finite-width kernels on the config-5 topology")
    ap.add_argument("--kruskal", action="store_true", help="--fw: random-Kruskal initial trees")
    a = ap.parse_args()
    if a.fw:
        return fw(a)
    ts, dims, out = synthetic.random_regular_tn(a.leaves, 3, 0)
    n_inds = 1 + max(i for xs in ts for i in xs)
    lm = ctree.pack_masks(ts, n_inds)
    seeds = np.arange(1, a.replicas + 1, dtype=np.uint32)
    links = core.random_trees(ts, n_inds, seeds)
    opt = core.BatchedOptimizer(lm, links, seeds, n_inds=n_inds)
    betas = np.linspace(0, 100, a.sweeps)
    opt.run(betas[:5])
    L = _lib.load()
    base = np.zeros(5, np.uint64)
    L.tnco_hip_diag_stage_cycles(opt._h, base.ctypes.data_as(C.c_void_p))
    m0 = opt.counters()["moves"]
    opt.run(betas)
    opt.sync()
    cyc = np.zeros(5, np.uint64)
    L.tnco_hip_diag_stage_cycles(opt._h, cyc.ctypes.data_as(C.c_void_p))
    cyc = (cyc - base).astype(np.float64)
    moves = opt.counters()["moves"] - m0
    it = cyc[4]
    names = ["mt19937", "state branches", "landing fence", "store phase"]
    if a.fine:  # library built with -DTNCO_PROFILE=2
        names = ["end-of-sweep blk", "addresses", "load issue", "move evaluation"]
    if opt.launch_groups == 0:  # few small trees: the LDS-resident kernel (csrc/sa_small.h)
        names = ["mt19937", "move", "sweep end", "sweep begin"]
        if a.fine:  # (-DTNCO_PROFILE=2: inside the move)
            names = ["move: operands", "move: costs", "move: acceptance", "move: update"]
    tot = cyc[:4].sum()
    print(f"replica-iterations {it:.3e}  moves {moves:.3e}  moves/iteration {moves / it:.3f}")
    for k in range(4):
        print(f"  {names[k]:16s} {cyc[k] / it:9.1f} cycles/iteration  {100 * cyc[k] / tot:5.1f} %")
    print(f"  total            {tot / it:9.1f} cycles/iteration (s_memtime ticks)")


def fw(a):
    ts, dims, out = synthetic.sycamore53_tn(20)
    n_inds = 1 + max(i for xs in ts for i in xs)
    lm = ctree.pack_masks(ts, n_inds)
    seeds = np.arange(1, a.replicas + 1, dtype=np.uint32)
    # (the reference's greedy starts, as bench.py; --kruskal: the random starts of round 1's line)
    links = core.random_trees(ts, n_inds, seeds) if a.kruskal else core.greedy_trees(ts, n_inds, seeds, device=0)
    opt = core.BatchedOptimizer(lm, links, seeds, n_inds=n_inds, max_width=40)
    L = _lib.load()
    base = np.zeros(5, np.uint64)
    L.tnco_hip_diag_stage_cycles(opt._h, base.ctypes.data_as(C.c_void_p))
    opt.run(np.linspace(0, 100, a.sweeps), update_slices_every=10)
    opt.sync()
    cyc = np.zeros(5, np.uint64)
    L.tnco_hip_diag_stage_cycles(opt._h, cyc.ctypes.data_as(C.c_void_p))
    cyc = (cyc - base).astype(np.float64)
    names = ["walk (post-order, widths)", "get_slices: too-wide counts", "get_slices: greedy pass", "rebuild + commit"]
    if a.fine:  # library built with -DTNCO_PROFILE=4: cycles inside the greedy pass
        names = ["scan for the next tensor that does not fit", "candidate positions", "shuffle", "keys + picks"]
    print(f"re-slices {cyc[4]:.3e} over {a.replicas} replicas (re-slicing sweeps only)")
    for k in range(4):
        print(f"  {names[k]:28s} {cyc[k] / cyc[4]:12.0f} cycles per re-slicing sweep  {100 * cyc[k] / cyc[:4].sum():5.1f} %")


if __name__ == "__main__":
    main()
