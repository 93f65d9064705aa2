"""method='sa' with a finite `max_width`: memory-constrained SA behind the plugin API.

Host mirror of /root/reference/tnco/app/finite_width/sa.py:73-289 (result type with
`disconnected_slices` / `slices`, `update_slices` keyword); the driver is
tnco_amd/app/_sa_driver.py, the kernels tnco_amd/csrc/fw_kernels.h.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import reduce
from typing import Any

from .._sa_driver import merge_contraction_paths, run_sa
from ..app import BaseContractionResults, BaseOptimizer

__all__ = ["Optimizer", "ContractionResults"]


@dataclass(repr=False, frozen=True, eq=False)
class ContractionResults(BaseContractionResults):
    """finite_width/sa.py:73-105."""
    disconnected_costs: list
    disconnected_paths: list
    disconnected_slices: list
    slices: frozenset
    _json_fields = BaseContractionResults._json_fields + ("disconnected_paths", "disconnected_slices", "slices")


class Optimizer(BaseOptimizer):
    """Simulated annealing under a maximum tensor width (index slicing), all runs on the GPU."""

    def optimize(self, tn: Any, betas, n_steps: int | None = None, n_runs: int = 1,
                 n_projs: int | None = None, timeout: float | None = None, update_slices: int = 10, *,
                 top_k: int | None = None, sweeps_per_launch: int = 100, prob: str = "mh",
                 device: int | None = None, initial_trees: str = "greedy", progress=None, **load_tn_options) -> Any:
        tn = self._load_tn(tn, **load_tn_options)
        if int(update_slices) != update_slices or update_slices <= 0:
            raise ValueError("'update_slices' must be a positive number.")
        merged, runtime = run_sa(self, tn, betas, n_steps, n_runs, n_projs, timeout, top_k=top_k,
                                 sweeps_per_launch=sweeps_per_launch, prob=prob, device=device,
                                 update_slices=int(update_slices), initial_trees=initial_trees, progress=progress)
        results = [ContractionResults(cost=c, runtime_s=runtime, path=mp,
                                      disconnected_costs=list(dc), disconnected_paths=paths,
                                      disconnected_slices=list(sl),
                                      slices=reduce(frozenset.union, sl, frozenset()))
                   for c, _gid, dc, paths, sl, mp in merged]
        return self._dump_results(tn, results)
