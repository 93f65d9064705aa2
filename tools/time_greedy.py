"""Initial trees as the reference draws them, host threads against the device kernels (diagnostic):
    python tools/time_greedy.py [n_leaves] [replicas]"""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import _lib, core, synthetic as syn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
prob = syn.sycamore_problem(20) if n == 0 else syn.regular_problem(n, 11)
seeds = syn.replica_seeds(R)
core.greedy_trees(prob.ts_inds, prob.n_inds, seeds[:64], device=0)  # (first call: module load)
t0 = time.perf_counter()
dev = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, device=0)
t1 = time.perf_counter()
redone = _lib.load().tnco_hip_diag_greedy_device_redone()
ns = min(R, 8192)
host = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds[:ns])
t2 = time.perf_counter()
print(f"{prob.n} tensors, {prob.n_inds} indices, {R} trees: device {t1 - t0:.3f} s ({redone} trees handed to the host); "
      f"host threads {(t2 - t1) * R / ns:.2f} s (from {ns} trees); equal: {np.array_equal(dev[:ns], host)}")
