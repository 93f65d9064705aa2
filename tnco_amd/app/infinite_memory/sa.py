"""method='sa', infinite memory: the SA optimizer behind the plugin API.

Host mirror of /root/reference/tnco/app/infinite_memory/sa.py:63-257; the driver itself is
tnco_amd/app/_sa_driver.py.  Differences a caller can observe:
  * the initial tree of a run is drawn as the reference draws it (Random(seed).shuffle of the tensors +
    opt_einsum's greedy, restated in csrc/host_greedy.cpp: opt_einsum is third-party and unpinned in
    the reference, so this stays "parity unpinned"); `initial_trees='kruskal'` selects the build's own
    random-Kruskal generator instead;
  * `load_tn` takes index lists only (no circuits, no arrays); pre-fusing (`fuse`, default 4) is this
    build's restatement of tnco/utils/tn.py:598-824;
  * only the `top_k` best runs (default min(n_runs, 1024)) are materialised as results.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

from .._sa_driver import expand_betas, merge_contraction_paths, run_sa  # noqa: F401 (re-exported)
from ..app import BaseContractionResults, BaseOptimizer, JSONEncoder  # noqa: F401 (re-exported)

__all__ = ["Optimizer", "ContractionResults", "merge_contraction_paths", "expand_betas"]


@dataclass(repr=False, frozen=True, eq=False)
class ContractionResults(BaseContractionResults):
    """sa.py:63-90: + per-component costs and paths (each path over ALL original tensors)."""
    disconnected_costs: list
    disconnected_paths: list
    _json_fields = BaseContractionResults._json_fields + ("disconnected_paths",)


class Optimizer(BaseOptimizer):
    """Simulated annealing with no memory constraint, all runs on the GPU."""

    def optimize(self, tn: Any, betas, n_steps: int | None = None, n_runs: int = 1,
                 n_projs: int | None = None, timeout: float | None = None, *, top_k: int | None = None,
                 sweeps_per_launch: int | None = None, prob: str = "mh", device: int | None = None,
                 initial_trees: str = "greedy", progress=None, **load_tn_options) -> Any:
        tn = self._load_tn(tn, **load_tn_options)
        merged, runtime = run_sa(self, tn, betas, n_steps, n_runs, n_projs, timeout, top_k=top_k,
                                 sweeps_per_launch=sweeps_per_launch, prob=prob, device=device,
                                 update_slices=None, initial_trees=initial_trees, progress=progress)
        results = [ContractionResults(cost=c, runtime_s=runtime, path=mp,
                                      disconnected_costs=list(dc), disconnected_paths=paths)
                   for c, _gid, dc, paths, _sl, mp in merged]
        return self._dump_results(tn, results)
