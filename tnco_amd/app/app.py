"""Front door: the reference's plugin API for this path.

Host mirror of /root/reference/tnco/app/app.py:64-94 (BaseContractionResults),
:573-712 (dump_results), :715-795 (BaseOptimizer) and :798-878 (the `Optimizer`
factory and its dispatch rule `tnco.app.{infinite_memory|finite_width}.<method>`).
"""
from __future__ import annotations

import bz2
import gzip
import io
import json
import pickle
from dataclasses import dataclass
from decimal import Decimal
from importlib import import_module
from pathlib import Path
from random import Random
from typing import Any

from .tn import TensorNetwork, load_tn

__all__ = ["Optimizer", "BaseOptimizer", "BaseContractionResults", "dump_results", "load_tn"]


class JSONEncoder(json.JSONEncoder):
    def default(self, obj):
        if isinstance(obj, Decimal):
            return str(obj)
        if isinstance(obj, BaseContractionResults):
            return dict(cost=obj.cost, runtime_s=obj.runtime_s, path=obj.path)
        if hasattr(obj, "to_json"):
            return obj.to_json()
        return super().default(obj)


@dataclass(repr=False, frozen=True, eq=False)
class BaseContractionResults:
    """cost / runtime_s / path (linear einsum format) -- app.py:64-94."""
    cost: float
    runtime_s: float
    path: list

    def __lt__(self, other):
        if not isinstance(other, BaseContractionResults):
            raise ValueError("Cannot compare against '{}'.".format(type(other).__name__))
        return self.cost < other.cost

    def __repr__(self):
        return "ContractionResults(cost={:1.3g}, runtime={:1.3g}s)".format(self.cost, self.runtime_s)

    def to_json(self):
        return json.dumps(self, cls=JSONEncoder)


def cost_to_decimal(x: float) -> Decimal:
    """The reference returns costs as Decimal(str) of the default-precision ostream print of the
    cost (6 significant digits; include/tnco/optimize/infinite_memory/optimizer.hpp:278-289,
    globals.hpp:48-53)."""
    return Decimal("%g" % x)


def dump_results(tn, res, *, output_format=None, output_filename=None, output_compression="auto",
                 overwrite_output_file=False, **kwargs):
    """(tn, res) | JSON string | file, as app.py:573-712."""
    check_only = kwargs.pop("check_only", False)
    if kwargs:
        raise TypeError("Unexpected extra keyword arguments.")
    output_format = "raw" if output_format is None else str(output_format).lower()
    if output_format not in ("raw", "json"):
        raise ValueError(f'"{output_format=}" not supported.')
    output_filename = None if output_filename is None else Path(output_filename).expanduser()
    if output_filename and not overwrite_output_file and output_filename.exists():
        raise FileExistsError("'{}' already exists. Please use 'overwrite_output_file=True'.".format(output_filename))
    output_compression = str(output_compression).lower()
    if output_compression not in ("auto", "none", "bz2", "gzip"):
        raise ValueError(f'"{output_compression=}" not supported.')
    if check_only:
        return None
    output = (tn, res)
    if output_format == "json":
        output = '{{"tn" : {}, "res" : {}}}'.format(tn.to_json(), "[" + ", ".join(r.to_json() for r in res) + "]")
    if output_filename:
        suffix = output_filename.suffix[1:] if output_compression == "auto" else output_compression
        open_, compress = (gzip.open, True) if suffix == "gzip" else (bz2.open, True) if suffix == "bz2" else (io.open, False)
        if isinstance(output, str):
            with open_(output_filename, "w") as f:
                f.write(output.encode() if compress else output)
            return None
        with open_(output_filename, "w" if compress else "bw") as f:
            pickle.dump(output, f)
        return None
    return output


@dataclass(frozen=True)
class BaseOptimizer:
    """Option record shared by the optimizers (app.py:715-795).  `n_jobs` is accepted for
    compatibility and ignored: replicas run on the GPU, not in worker processes."""
    max_width: float | None = None
    n_jobs: int = -1
    width_type: str = "float32"
    cost_type: str = "float64"
    output_format: str | None = None
    output_filename: str | None = None
    output_compression: str = "auto"
    overwrite_output_file: bool = False
    atol: float = 1e-5
    dtype: Any | None = None
    backend: str | None = None
    seed: int | None = None
    verbose: int = False

    def optimize(self, *args, **kwargs):
        raise NotImplementedError()

    def _load_tn(self, tn, **opts) -> TensorNetwork:
        return load_tn(tn, atol=self.atol, dtype=self.dtype, backend=self.backend, seed=self.seed,
                       verbose=self.verbose, **opts)

    def _dump_results(self, tn, res, **opts):
        return dump_results(tn, res, output_format=self.output_format, output_filename=self.output_filename,
                            output_compression=self.output_compression,
                            overwrite_output_file=self.overwrite_output_file, **opts)

    def __post_init__(self):
        object.__setattr__(self, "_rng", Random(self.seed))
        self._dump_results(None, None, check_only=True)


def Optimizer(method: str = "sa", max_width: float | None = None, n_jobs: int = -1,
              width_type: str = "float32", cost_type: str = "float64", output_format: str | None = None,
              output_filename: str | None = None, output_compression: str = "auto",
              overwrite_output_file: bool = False, atol: float = 1e-5, dtype: Any | None = None,
              backend: str | None = None, seed: int | None = None, verbose: int = False) -> BaseOptimizer:
    """Factory with the reference's dispatch rule (app.py:866-878): finite `max_width` selects
    `<pkg>.app.finite_width.<method>`, otherwise `<pkg>.app.infinite_memory.<method>`."""
    opts = dict(locals())
    opts.pop("method")
    module = __name__.rsplit(".", 1)[0]
    module += ".finite_width" if (max_width is not None and max_width < float("inf")) else ".infinite_memory"
    module += "." + str(method)
    return import_module(module).Optimizer(**opts)
