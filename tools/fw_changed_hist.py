"""How many indices does a re-slice change?  (diagnostic: the re-pricing of fw_wave_kernel handles FWT_MAXD = 64 /
128; more -- or an index it cannot follow -- is a full rebuild by fw_reslice_b_kernel, which the whole batch waits for)
    python tools/fw_changed_hist.py [max_width] [replicas]"""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import core, synthetic as syn  # noqa: E402

mw = float(sys.argv[1]) if len(sys.argv) > 1 else 32
R = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
prob = syn.sycamore_problem(20)
seeds = syn.replica_seeds(R)
links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0, keep_on_device=True)
opt = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds, max_width=mw, width_type="float32")
betas = syn.linear_betas(0, 100, 1200)
hist = np.zeros(200, np.int64)
unknown = 0
zero_by_quarter = np.zeros((4, 2), np.int64)  # (proposal == current slices, all) per quarter of the schedule
for s in range(0, 1200, 10):
    opt.run(betas[s:s + 10], update_slices_every=10)
    how, nch = opt.reslice_info()
    unknown += int((nch < 0).sum())
    hist += np.bincount(np.clip(nch[nch >= 0], 0, 199), minlength=200)
    zero_by_quarter[s // 300] += (int((nch == 0).sum()), int((nch >= 0).sum()))
    if s % 300 == 290:
        # exactness of the partial sums: every contraction cost is 2^e, so every sum is exact while
        # log2(total) - min e < 53 (all addends are multiples of 2^min e)
        span = []
        for r in range(0, R, max(1, R // 48)):
            cc, pc, _hy = opt.caches(r)
            e = np.log2(cc[prob.n:])
            span.append((float(np.log2(pc[-1])) - float(e.min()), int(bin(int.from_bytes(opt.slices(r)[0].tobytes(), "little")).count("1"))))
        print(f"sweep {s + 10}: log2(total) - min log2(contraction cost) over 48 replicas: max {max(x for x, _ in span):.1f}, "
              f"median {np.median([x for x, _ in span]):.1f}; sliced indices median {int(np.median([k for _, k in span]))}", flush=True)
    if s % 300 == 0:
        big = nch[nch > 48]
        print(f"sweep {s + 10}: rebuilt in full {int((how == 0).sum())}, changed > 48: {len(big)} {sorted(big.tolist())[-8:]}, "
              f"median {int(np.median(nch[nch >= 0]))}", flush=True)
tot = hist.sum()
cum = np.cumsum(hist) / tot
for q in range(4):
    print(f"  sweeps {300 * q + 1}-{300 * q + 300}: proposal == current slices in {zero_by_quarter[q, 0] / max(1, zero_by_quarter[q, 1]):.3f} of the re-slices")
print("replica re-slices", int(tot), "unknown", unknown, "stats", opt.fw_stats())
for q in (0.5, 0.9, 0.99, 0.999, 0.9999, 0.99999):
    print(f"  quantile {q}: {int(np.searchsorted(cum, q))} changed indices")
print("  > 32:", int(hist[33:].sum()), " > 64:", int(hist[65:].sum()), " > 96:", int(hist[97:].sum()), " max:", int(np.nonzero(hist)[0].max()))
