"""Memory-constrained SA (finite `max_width`): not built yet.

The reference's twin of the hot path (include/tnco/optimize/finite_width/greedy/optimizer.hpp:117-390,
driven by tnco/app/finite_width/sa.py) is restated in the oracle (oracle/tnco_oracle.c: update_fw)
but has no HIP kernel in this round.  There is no CPU fallback by design.
"""
from ..app import BaseOptimizer


class Optimizer(BaseOptimizer):
    def optimize(self, *args, **kwargs):
        raise NotImplementedError("method='sa' with a finite max_width has no GPU kernel yet "
                                  "(SURVEY.md section 8 row a13).")
