"""End-to-end wall time of Optimizer.optimize() (diagnostic): the README example, a mid-size batch and
BASELINE config 3 through the plugin API, with and without the default pre-fusing."""
import pathlib
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import synthetic as syn  # noqa: E402
from tnco_amd.app import Optimizer  # noqa: E402

warnings.simplefilter("ignore")


def spec_of(n, seed):
    ts, _d, _ = syn.random_regular_tn(n, 3, seed)
    n_inds = max(max(x) for x in ts) + 1
    return [(2, *[f"t{t}" for t in range(n) if k in ts[t]]) for k in range(n_inds)]


def timed(label, spec, **kw):
    opt = Optimizer(method="sa", seed=0)
    t0 = time.perf_counter()
    tn, res = opt.optimize(spec, betas=(0, 100), **kw)
    dt = time.perf_counter() - t0
    c = float(res[0].cost)
    print(f"{label:46s} {dt:7.2f} s wall   {len(tn.tensors):4d} tensors after load   "
          f"best log2(cost) {np.log2(c) if c > 0 else float('-inf'):.3f}")


timed("README chain, 8 runs x 100 steps (warm-up)", "2 a b\n2 b c\n2 c d", n_steps=100, n_runs=8, fuse=None)
timed("README chain, 8 runs x 100 steps", "2 a b\n2 b c\n2 c d", n_steps=100, n_runs=8, fuse=None)
timed("README chain, default fuse", "2 a b\n2 b c\n2 c d", n_steps=100, n_runs=8)
timed("64 leaves, 4096 runs x 1000 steps, fuse=None", spec_of(64, 7), n_steps=1000, n_runs=4096, fuse=None)
timed("512 leaves, 65536 runs x 1000 steps, fuse=None", spec_of(512, 11), n_steps=1000, n_runs=65536, top_k=16, fuse=None)
timed("512 leaves, 65536 runs x 1000 steps, fuse=4", spec_of(512, 11), n_steps=1000, n_runs=65536, top_k=16)
