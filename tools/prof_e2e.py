"""cProfile of one optimize() call of 65536 runs with the default head of 1024 results."""
import cProfile, pstats, sys, warnings
sys.path.insert(0, '.')
warnings.simplefilter("ignore")
from tnco_amd import synthetic as syn
from tnco_amd.app import Optimizer
ts, _d, _ = syn.random_regular_tn(512, 3, 11)
spec = [(2, *[f"t{t}" for t in range(512) if k in ts[t]]) for k in range(768)]
Optimizer(method="sa", seed=0).optimize("2 a b\n2 b c\n2 c d", betas=(0, 100), n_steps=10, n_runs=8, fuse=None)
pr = cProfile.Profile(); pr.enable()
Optimizer(method="sa", seed=0).optimize(spec, betas=(0, 100), n_steps=1000, n_runs=65536, fuse=None, initial_trees="kruskal")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
