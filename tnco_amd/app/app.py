"""Front door: the reference's plugin API for this path.

Host mirror of /root/reference/tnco/app/app.py:64-94 (BaseContractionResults),
:573-712 (dump_results), :715-795 (BaseOptimizer) and :798-878 (the `Optimizer`
factory and its dispatch rule `tnco.app.{infinite_memory|finite_width}.<method>`).
"""
from __future__ import annotations

import bz2
import gzip
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
    """Results as JSON objects of the fields their class lists in `_json_fields` (the keys the reference's
    encoders emit: app.py:47-61, infinite_memory/sa.py:47-60, finite_width/sa.py:42-70), costs as strings,
    index sets as sorted lists."""

    def default(self, obj):
        fields = getattr(type(obj), "_json_fields", None)
        if fields is not None:
            return {name: getattr(obj, name) for name in fields}
        if isinstance(obj, Decimal):
            return str(obj)
        if isinstance(obj, (set, frozenset)):
            return sorted(obj, key=str)
        to_json = getattr(obj, "to_json", None)
        return to_json() if callable(to_json) else super().default(obj)


@dataclass(repr=False, frozen=True, eq=False)
class BaseContractionResults:
    """cost / runtime_s / path (linear einsum format) -- app.py:64-94."""
    cost: float
    runtime_s: float
    path: list
    _json_fields = ("cost", "runtime_s", "path")

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


_FORMATS = ("raw", "json")
_COMPRESSIONS = {"auto": None, "none": None, "bz2": bz2.compress, "gzip": gzip.compress}


def _choice(option: str, value, allowed) -> str:
    """The lower-cased option value, or the reference's ValueError text for one it does not know."""
    text = str(value).lower()
    if text not in allowed:
        raise ValueError(f'"{option}={text!r}" not supported.')
    return text


def _file_bytes(tn, res, as_json: bool) -> bytes:
    """What a results file holds before compression: the JSON text, or the pickle of (tn, res)."""
    return _json_text(tn, res).encode() if as_json else pickle.dumps((tn, res))


def _json_text(tn, res) -> str:
    return '{"tn" : ' + tn.to_json() + ', "res" : [' + ", ".join(r.to_json() for r in res) + "]}"


def dump_results(tn, res, *, output_format=None, output_filename=None, output_compression="auto",
                 overwrite_output_file=False, check_only=False, **unexpected):
    """Hand the results back, or put them in a file (contract: tnco/app/app.py:573-637).

    No file name: returns (tn, res) for `output_format` None / 'raw', the JSON text for 'json'.  With a file name:
    writes the JSON text, or the pickle of (tn, res), compressed as `output_compression` says ('auto': by the
    file's suffix, .gzip / .bz2), and returns None.  `check_only` validates the options and does nothing else.

    One serialisation to bytes, one optional compressor, one write -- the compressed files read back with
    gzip.open / bz2.open exactly as the reference's do."""
    if unexpected:
        raise TypeError("Unexpected extra keyword arguments.")
    fmt = _choice("output_format", "raw" if output_format is None else output_format, _FORMATS)
    scheme = _choice("output_compression", output_compression, _COMPRESSIONS)
    target = Path(output_filename).expanduser() if output_filename is not None and str(output_filename) else None
    if target is not None and target.exists() and not overwrite_output_file:
        raise FileExistsError(f"'{target}' already exists. Please use 'overwrite_output_file=True'.")
    if check_only:
        return None
    as_json = fmt == "json"
    if target is None:
        return _json_text(tn, res) if as_json else (tn, res)
    if scheme == "auto":
        scheme = target.suffix.lstrip(".")
    squeeze = _COMPRESSIONS.get(scheme)
    blob = _file_bytes(tn, res, as_json)
    target.write_bytes(squeeze(blob) if squeeze else blob)
    return None


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
