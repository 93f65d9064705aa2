"""Where does create() spend its time?  (first handle of a fresh process; TNCO_HIP_JLOG_CAP sets the log size)"""
import sys, time, warnings
sys.path.insert(0, '.')
warnings.simplefilter("ignore")
import numpy as np
from tnco_amd import core, synthetic as syn
prob = syn.regular_problem(512, 11)
seeds = syn.replica_seeds(65536)
links = core.random_trees(prob.ts_inds, prob.n_inds, seeds)
for k in range(2):
    t0 = time.perf_counter()
    opt = core.BatchedOptimizer(prob.leaf_masks, links, seeds, n_inds=prob.n_inds); opt.sync()
    t1 = time.perf_counter()
    opt.run(syn.linear_betas(0, 100, 100)); opt.sync()
    t2 = time.perf_counter()
    opt.close()
    t3 = time.perf_counter()
    print(f"handle {k}: create {t1 - t0:.3f} s  (device bytes {opt.device_bytes / 2**30 if opt._h else 0:.1f}), 100 sweeps {t2 - t1:.3f} s, destroy {t3 - t2:.3f} s")
