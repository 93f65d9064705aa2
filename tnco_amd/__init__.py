"""tnco_amd -- MI355X-native simulated annealing of tensor-network contraction trees.

One hot path of google-research/tnco (the SA inner loop behind
`tnco.app.Optimizer(method='sa').optimize(tn, betas, n_steps, n_runs)`), rebuilt as hand-written
HIP kernels for gfx950 behind a C ABI (include/tnco_hip.h).  There is no CPU fallback: importing
`tnco_amd.core` without the built library fails.
"""
from .app import Optimizer  # noqa: F401

__version__ = "0.5"  # (= the library: tnco_hip_version)
