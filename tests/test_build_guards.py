"""The compiler-dependent properties the measured numbers lean on, asserted on the gfx950 code objects inside the
libtnco_hip.so of the tree (tools/code_objects.py; no GPU, a few seconds): register budgets -> wavefronts per SIMD,
scratch, and the shape of the sweep kernels' main loop -- ONE landing fence ("the single
s_waitcnt vmcnt of the loop"), no scratch access inside it.  A toolchain bump that breaks one of them fails here, not
in a benchmark."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def report():
    import code_objects
    if not code_objects.LIB.exists():
        pytest.fail("tnco_amd/libtnco_hip.so is missing: run __graft_entry__.build()")
    if not (code_objects.LLVM / "llvm-objdump").exists():
        pytest.skip("no ROCm LLVM tools on this machine")
    return code_objects.report()


def test_register_budgets_of_the_occupancy_critical_kernels(report):
    rows, _ = report
    assert len(rows) == 4
    for ok, line in rows:
        assert ok, line


def test_sweep_loops_have_one_landing_fence_and_no_scratch(report):
    """Also the hyper-index and the general-cost-model instantiations: round 5 found 102 `s_waitcnt vmcnt(0)` in the
    latter's loop (tables read from device memory inside it) -- 15-40 % of its throughput."""
    _, staged = report
    assert len(staged) == 5
    for what, rep in staged.items():
        assert rep["fences"] == 1, (what, rep["vm_waits"], rep["memory_between_waits"])
        import code_objects
        assert code_objects.scratch_ok(what, rep), (what, rep["scratch_in_loop"])
        # (the loop of the state machine: one load sequence, one store sequence -- a duplicated body would double these)
        assert rep["loads"] <= 20 and 5 <= rep["stores"] <= 16, (what, rep["loads"], rep["stores"])


def test_waves_per_simd_rule():
    import code_objects
    assert [code_objects.waves_per_simd(v) for v in (64, 96, 127, 128, 129, 168, 169, 215, 256, 257)] == [8, 5, 4, 4, 3, 3, 2, 2, 2, 1]


def test_lds_budget_of_the_small_tree_kernels(report):
    """csrc/sa_small.h: the host decides from small_replicas_per_cu() (64 up to 64 leaves, 32 beyond) whether every replica
    of a handle is LDS-resident at once -- that is the kernels' static LDS (38.5 and 72.5 KiB per 16-replica block: four and
    two blocks in a CU's 160 KiB), which only the code object knows; and neither kernel may use scratch."""
    import code_objects
    seen = {}
    for elf in code_objects.code_objects():
        for name, meta in code_objects.kernel_table(elf).items():
            if "sa_small_kernel" in name:
                seen[63 if "ILi63E" in name else 127] = meta
    assert sorted(seen) == [63, 127]
    for ni, per_cu in ((63, 64), (127, 32)):
        lds = seen[ni]["group_segment_fixed_size"]
        assert (160 * 1024 // lds) * 16 == per_cu, (ni, lds)
        assert seen[ni]["private_segment_fixed_size"] == 0 and seen[ni]["vgpr_count"] <= 256


def test_the_lds_kernels_of_larger_trees_use_no_scratch(report):
    """sa_lds_kernel<K, HYPER> (csrc/sa_small.h): one wavefront per block and at most four blocks per CU -- registers do not
    bound its occupancy (LDS does, and the host carves that), scratch would: none, in all eight instantiations."""
    import code_objects
    seen = {}
    for elf in code_objects.code_objects():
        for name, meta in code_objects.kernel_table(elf).items():
            if "sa_lds_kernel" in name:
                seen[name] = meta
    assert len(seen) == 8, sorted(seen)
    for name, meta in seen.items():
        assert meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0 and meta["vgpr_count"] <= 256, (name, meta)
