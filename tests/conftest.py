import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


# PyTorch-ROCm's bundled HIP runtime must initialise before libtnco_hip.so's (tnco_amd/_lib.py): the -m gpu
# tests that use torch (device-side reduction operand, two ranks) share the session with tests that do not.
try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001 -- no torch: nothing to order
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as orc
    orc.build()
    return orc
