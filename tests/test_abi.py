"""The C-ABI library loads and exports every symbol include/tnco_hip.h declares (no compute)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "tnco_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tnco_hip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tnco_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"libtnco_hip.so does not export {name}"
    assert sorted(_lib.EXPORTS) == names
    assert lib.tnco_hip_version().startswith(b"tnco_hip")


def test_desc_struct_matches_header():
    """ctypes mirror of tnco_hip_desc: same field order as the header."""
    from tnco_amd import _lib
    txt = (ROOT / "include" / "tnco_hip.h").read_text()
    body = re.search(r"typedef struct tnco_hip_desc \{(.*?)\} tnco_hip_desc;", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [re.findall(r"(\w+)\s*;", ln)[0] for ln in body.split("\n") if ";" in ln]
    assert fields == [f[0] for f in _lib.Desc._fields_]
    assert ctypes.sizeof(_lib.Desc) == 192


def test_integration_stub_matches_header():
    """The reference-side ctypes stub shown in INTEGRATION.md declares the descriptor with the fields
    of the header, in its order."""
    from tnco_amd import _lib
    md = (ROOT / "INTEGRATION.md").read_text()
    stub = md[md.index("class Desc(C.Structure)"):md.index("_lib.tnco_hip_create.argtypes")]
    assert re.findall(r'\("(\w+)", C\.', stub) == [f[0] for f in _lib.Desc._fields_]


def test_error_reporting_without_gpu():
    """Argument errors are reported through the status code + last_error, before any device use."""
    from tnco_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.tnco_hip_create(None, ctypes.byref(h)) == _lib.EINVAL
    assert b"null" in lib.tnco_hip_last_error()
    assert lib.tnco_hip_sync(None) == _lib.EINVAL
    d = _lib.Desc()
    d.n_leaves, d.n_replicas = 1, 1
    assert lib.tnco_hip_create(ctypes.byref(d), ctypes.byref(h)) == _lib.EINVAL
    assert lib.tnco_hip_last_error() == b"Precision is too low."
