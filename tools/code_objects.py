"""What the compiler made of the kernels that ship: the gfx950 code objects inside libtnco_hip.so (no GPU needed).

    python tools/code_objects.py              # budgets + the sweep loop's wait / scratch check, a few seconds

The measured numbers lean on three things the source cannot promise: the register budgets that set
the wavefronts per SIMD (168 VGPRs -> 3 for the headline kernel; fw_wave_kernel lost 20 % twice at 129 instead of 127),
the ONE landing fence of the sweep loop (a second `s_waitcnt vmcnt` in the loop body serialises the walk of sixteen
replicas), and no scratch access inside that loop.  tests/test_build_guards.py asserts them on the library in the tree:
a ROCm bump fails a test, not a benchmark.  (tools/check_registers.py re-compiles two translation units for the same
register numbers: minutes.)
"""
from __future__ import annotations

import pathlib
import re
import struct
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parents[1]
LIB = ROOT / "tnco_amd" / "libtnco_hip.so"
LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# mangled-name fragment -> (what, max VGPRs, min wavefronts per SIMD, max scratch bytes per lane)
BUDGET = {
    "sa_run_kernelILi2ELi3ELb0ELb0ELb0E": ("sa_run_kernel<2, 3, false, false, false> (headline leg)", 168, 3, 16),
    "sa_run_fw_kernelILi2ELi4ELb0E": ("sa_run_fw_kernel<2, 4, false> (finite-width moves: 192 VGPRs leave a SIMD room for a fw_wave_kernel wavefront)", 192, 2, 48),
    "fw_wave_kernelILi9ELi4ELb0ELb0E": ("fw_wave_kernel<9, 4, false, false> (config 5 re-slice)", 128, 4, 0),
    "fw_wave_kernelILi6ELi4ELb1ELb0E": ("fw_wave_kernel<6, 4, true, false> (hyper-index networks up to 384 nodes)", 128, 4, 0),
}
# kernels whose main loop must hold one landing fence and no scratch access (the last two: not in BUDGET, loop check only)
STAGED = ("sa_run_kernelILi2ELi3ELb0ELb0ELb0E", "sa_run_fw_kernelILi2ELi4ELb0E")
STAGED_ONLY = {
    "sa_run_kernelILi2ELi3ELb1ELb0ELb0E": "sa_run_kernel<2, 3, true, false, false> (hyper-indices)",
    "sa_run_kernelILi2ELi3ELb0ELb1ELb0E": "sa_run_kernel<2, 3, false, true, false> (general cost models: tables in LDS)",
    "sa_run_kernelILi2ELi3ELb1ELb1ELb0E": "sa_run_kernel<2, 3, true, true, false> (hyper-indices + general cost models)",
}


def code_objects(lib: pathlib.Path = LIB) -> list[bytes]:
    """The gfx950 ELF images of every offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = pathlib.Path(td) / "fatbin"
        subprocess.check_call([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)])
        d = fat.read_bytes()
    return bundle_objects(d)


def bundle_objects(d: bytes) -> list[bytes]:
    """The gfx950 ELF images inside clang offload bundles found in `d` (a .hip_fatbin section, or a --cuda-device-only object)."""
    out = []
    for m in re.finditer(MAGIC, d):
        o = m.start()
        n = struct.unpack_from("<Q", d, o + 24)[0]
        p = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                out.append(d[o + off:o + off + size])
    return out


def kernel_table(elf: bytes) -> dict[str, dict]:
    """{mangled kernel name: {vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, private_segment_fixed_size,
    group_segment_fixed_size}} from the code object's metadata note."""
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
    table = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))  # noqa: E731
        table[name] = {k: get(k) for k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count",
                                           "private_segment_fixed_size", "group_segment_fixed_size")}
    return table


def short_kernel_name(demangled: str) -> str:
    """`void tnco::sa_run_kernel<2, 3, false>(tnco::Params, ...)` -> `sa_run_kernel<2, 3, false>`: what a rocprofv3 trace
    row and a code object's symbol have in common."""
    s = demangled.strip().strip('"')
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):  # (the argument list opens at the first "(" outside the template brackets)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not s.startswith("(anonymous namespace)", i):
            cut = i
            break
    s = s[:cut]
    for junk in ("void ", "tnco::", "(anonymous namespace)::"):
        s = s.replace(junk, "")
    return s.replace(".kd", "").strip()


def kernel_names(lib: pathlib.Path = LIB) -> set[str]:
    """Short names of every gfx950 kernel the library ships."""
    mangled = set()
    for elf in code_objects(lib):
        mangled.update(kernel_table(elf))
    import shutil
    filt = shutil.which("c++filt") or str(LLVM / "llvm-cxxfilt")
    out = subprocess.run([filt], input="\n".join(sorted(mangled)), capture_output=True, text=True).stdout
    return {short_kernel_name(ln) for ln in out.splitlines() if ln.strip()}


def waves_per_simd(vgprs: int) -> int:
    """gfx950: 512 registers per lane and SIMD (VGPRs + AGPRs, one file), allocated in blocks of 8, at most 8 wavefronts."""
    return min(8, 512 // max(8, -(-vgprs // 8) * 8))


def disassemble(elf: bytes, symbol: str) -> list[tuple[int, str, str]]:
    """[(address, mnemonic, operands)] of one kernel."""
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run([str(LLVM / "llvm-objdump"), "-d", f"--disassemble-symbols={symbol}", f.name],
                             capture_output=True, text=True).stdout
    ins = []
    for ln in txt.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def loops(ins):
    """(head, tail) address pairs of the backward branches."""
    out = []
    for a, op, args in ins:
        if op == "s_branch" or op.startswith("s_cbranch"):
            o = int(args)
            o = o - 65536 if o >= 32768 else o
            t = a + 4 + 4 * o
            if t <= a:
                out.append((t, a))
    return out


def main_loop_report(ins) -> dict:
    """The kernel's main loop = the smallest backward branch that spans the staged load sequence and the store phase (at least
    8 global loads and 5 global stores); the widest one if none does.  Inside it, outside the loops nested in it (the rare
    paths: the full copy of the best tree; the cost models' leg loops): the `s_waitcnt vmcnt` instructions, whether loads
    sit between the first and the last of them (more than one landing fence), and scratch accesses."""
    lp = loops(ins)
    is_ld = lambda op: op.startswith(("global_load", "buffer_load"))  # noqa: E731
    is_st = lambda op: op.startswith(("global_store", "buffer_store"))  # noqa: E731

    def counts(t, a):
        b = [i for i in ins if t <= i[0] <= a]
        return sum(1 for i in b if is_ld(i[1])), sum(1 for i in b if is_st(i[1]))

    good = [(a - t, t, a) for t, a in lp if counts(t, a)[0] >= 8 and counts(t, a)[1] >= 5]
    if good:
        _, head, tail = min(good)
        # (several back edges to one head: the loop is the widest of them)
        tail = max(a for t, a in lp if t == head or (head <= t and a >= tail and t - head < 64))
    else:
        head, tail = max(lp, key=lambda x: x[1] - x[0])
    nested = [(t, a) for t, a in lp if head < t and a < tail]
    inner = lambda x: any(t <= x <= a for t, a in nested if (a - t) < (tail - head) * 0.5)  # noqa: E731
    body = [i for i in ins if head <= i[0] <= tail]
    flat = [i for i in body if not inner(i[0])]
    is_mem = lambda op: op.startswith(("global_", "buffer_", "flat_", "scratch_"))  # noqa: E731
    waits = [i for i in flat if i[1] == "s_waitcnt" and "vmcnt" in i[2]]
    between = [i for i in flat if waits and waits[0][0] < i[0] < waits[-1][0] and is_mem(i[1])]
    return {"instructions": len(body), "vm_waits": [(hex(a - ins[0][0]), args) for a, _op, args in waits],
            "fences": 0 if not waits else 1 + sum(1 for i in between if i[1].startswith(("global_load", "buffer_load", "flat_load"))),
            "memory_between_waits": [(hex(a - ins[0][0]), op) for a, op, _ in between],
            "scratch_in_loop": [(hex(a - ins[0][0]), op) for a, op, _ in body if op.startswith("scratch_")],
            "loads": sum(1 for i in flat if is_ld(i[1])), "stores": sum(1 for i in flat if is_st(i[1]))}


def report(lib: pathlib.Path = LIB):
    rows, staged = [], {}
    objs = code_objects(lib)
    found = {}
    for elf in objs:
        for name, meta in kernel_table(elf).items():
            for frag in list(BUDGET) + list(STAGED_ONLY):
                if frag in name:
                    found[frag] = (name, meta, elf)
    for frag, (what, max_v, min_w, max_scr) in BUDGET.items():
        if frag not in found:
            rows.append((False, f"{what}: not in {lib.name}"))
            continue
        name, meta, elf = found[frag]
        v, w, scr = meta["vgpr_count"], waves_per_simd(meta["vgpr_count"]), meta["private_segment_fixed_size"]
        ok = v <= max_v and w >= min_w and scr <= max_scr
        rows.append((ok, f"{what}: {v} VGPRs (budget {max_v}), {w} wavefronts per SIMD (at least {min_w}), scratch {scr} B/lane "
                         f"(at most {max_scr}), {meta['vgpr_spill_count']} VGPRs spilled, LDS {meta['group_segment_fixed_size']} B"))
        if frag in STAGED:
            staged[what] = main_loop_report(disassemble(elf, name))
    for frag, what in STAGED_ONLY.items():
        if frag in found:
            name, _meta, elf = found[frag]
            staged[what] = main_loop_report(disassemble(elf, name))
        else:
            staged[what] = {"instructions": 0, "vm_waits": [], "fences": 0, "memory_between_waits": [], "scratch_in_loop": ["kernel not found"],
                            "loads": 0, "stores": 0}
    return rows, staged


def scratch_ok(what: str, rep: dict) -> bool:
    """No scratch access inside a sweep loop -- but for the finite-width moves under their 192-register ceiling: at most one
    spill and one reload ahead of the landing fence, the forms that were measured (profiles/experiments_r06.md)."""
    acc = rep["scratch_in_loop"]
    if what.startswith("sa_run_fw_kernel"):
        return len(acc) <= 2 and sum(1 for _a, op in acc if op.startswith("scratch_store")) <= 1
    return not acc


def main():
    rows, staged = report()
    bad = 0
    for ok, line in rows:
        bad += not ok
        print(("ok   " if ok else "OVER ") + line)
    for what, rep in staged.items():
        ok = rep["fences"] == 1 and scratch_ok(what, rep)
        bad += not ok
        print(("ok   " if ok else "BAD  ") + f"{what}: main loop of {rep['instructions']} instructions, {rep['loads']} loads / "
              f"{rep['stores']} stores, landing fences {rep['fences']} (vmcnt waits {rep['vm_waits']}), scratch accesses in the loop "
              f"{rep['scratch_in_loop']}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
