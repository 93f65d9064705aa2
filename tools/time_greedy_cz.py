import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import numpy as np
from tnco_amd import _lib, core, synthetic as syn
for depth, fuse in ((20, 4), (12, None)):
    ts, dims, out = syn.sycamore53_cz_tn(depth=depth, fuse=fuse)[:3]
    n_inds = max(max(x) for x in ts if x) + 1
    seeds = syn.replica_seeds(65536)
    from tnco_amd import ctree as ct
    cnt = {}
    for xs in ts:
        for i in xs: cnt[i] = cnt.get(i, 0) + 1
    keep = [i for i in out if cnt.get(i, 0) <= 1]
    om = ct.pack_masks([keep], n_inds)[0]
    core.greedy_trees(ts, n_inds, seeds[:64], output_mask=om, device=0)
    t0 = time.perf_counter()
    core.greedy_trees(ts, n_inds, seeds, output_mask=om, device=0, keep_on_device=True)
    print(f"cz depth {depth} fuse {fuse}: {len(ts)} tensors, {n_inds} indices, max holders {max(cnt.values())}: {time.perf_counter()-t0:.3f} s, redone {_lib.load().tnco_hip_diag_greedy_device_redone()}")
