"""profiles/ must describe the library that ships: every kernel named in profiles/pmc_traffic.json (`trace_names`) and
in the newest profiles/*_kernel_stats.csv has to exist, template arguments and all, among the gfx950 kernels inside
tnco_amd/libtnco_hip.so (tools/code_objects.py; no GPU).  A kernel commit that changes an instantiation without a
re-run of the profile set fails here."""
import csv
import json
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def shipped():
    import code_objects
    if not code_objects.LIB.exists():
        pytest.fail("tnco_amd/libtnco_hip.so is missing: run __graft_entry__.build()")
    import shutil
    if not (code_objects.LLVM / "llvm-readelf").exists() or not (shutil.which("c++filt") or (code_objects.LLVM / "llvm-cxxfilt").exists()):
        pytest.skip("no ROCm LLVM tools / demangler on this machine")
    names = code_objects.kernel_names()
    assert any(n.startswith("sa_run_kernel<") for n in names) and any(n.startswith("fw_wave_kernel<") for n in names)
    return names


def newest_kernel_stats() -> Path:
    files = sorted(ROOT.glob("profiles/r*_kernel_stats.csv"), key=lambda f: (int(re.match(r"r(\d+)", f.name).group(1)), f.name))
    return files[-1]


def test_short_kernel_name():
    import code_objects
    f = code_objects.short_kernel_name
    assert f('"void tnco::sa_run_kernel<2, 3, false, false, false>(tnco::Params, double const*, long, int, tnco::FwParams, int, int)"') \
        == "sa_run_kernel<2, 3, false, false, false>"
    assert f("tnco::materialize_min_kernel(tnco::Params, tnco::Links*, long, long)") == "materialize_min_kernel"
    assert f("void tnco::(anonymous namespace)::greedy_graph_kernel<4>(int)") == "greedy_graph_kernel<4>"


def test_pmc_traffic_names_kernels_of_the_shipped_library(shipped):
    import code_objects
    j = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    seen = [n for group in j["trace_names"].values() for n in group]
    assert seen
    stale = [n for n in seen if code_objects.short_kernel_name(n) not in shipped]
    assert not stale, f"profiles/pmc_traffic.json was taken from another library: {stale}"


def test_newest_kernel_stats_names_kernels_of_the_shipped_library(shipped):
    import code_objects
    f = newest_kernel_stats()
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    ours = [r["Name"] for r in rows if "tnco::" in r["Name"]]
    assert any("sa_run_kernel" in n for n in ours), f
    stale = [n for n in ours if code_objects.short_kernel_name(n) not in shipped]
    assert not stale, f"{f.name} was taken from another library: {stale}"
