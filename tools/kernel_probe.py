"""Compile ONE instantiation of the sweep kernel for gfx950 (seconds) and report its registers, spills and the shape of its main
loop -- the edit/compile loop for register-pressure work (no GPU).

    python tools/kernel_probe.py "2, 4, false, false, true, false" [-DTNCO_FW_STAGED_WAVES=3 ...]
"""
import pathlib
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import code_objects as co  # noqa: E402


def main():
    targs = sys.argv[1] if len(sys.argv) > 1 else "2, 4, false, false, true, false"
    flags = sys.argv[2:]
    kernel = "sa_run_kernel"
    if targs.startswith("fw:"):  # "fw:2, 4, false" = sa_run_fw_kernel<2, 4, false>
        kernel, targs = "sa_run_fw_kernel", targs[3:]
    with tempfile.TemporaryDirectory() as td:
        src = pathlib.Path(td) / "probe.hip"
        src.write_text('#include "sa_sweep.h"\nnamespace tnco {\ntemplate __global__ void %s<%s>(const Params, const double* __restrict__, '
                       'const int64_t, const int, const FwParams, const int, const int);\n}\n' % (kernel, targs))
        obj = pathlib.Path(td) / "probe.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                               "-Wno-unused-function", "-I", str(ROOT / "tnco_amd" / "csrc"), "--cuda-device-only", "-c", "-o", str(obj), str(src), *flags])
        raw = obj.read_bytes()
        elf = raw if raw[:4] == b"\x7fELF" else co.bundle_objects(raw)[0]
        for name, meta in co.kernel_table(elf).items():
            if kernel not in name:
                continue
            ins = co.disassemble(elf, name)
            rep = co.main_loop_report(ins)
            t, a = max(co.loops(ins), key=lambda x: x[1] - x[0])  # the outer loop: the widest backward branch
            outer = [i for i in ins if t <= i[0] <= a]
            kinds = {"VALU": "v_", "SALU": "s_", "LDS": "ds_", "VMEM": ("global_", "buffer_", "scratch_")}
            mix = {k: sum(1 for i in outer if i[1].startswith(v)) for k, v in kinds.items()}
            print(f"VGPRs {meta['vgpr_count']}  spilled {meta['vgpr_spill_count']}  scratch {meta['private_segment_fixed_size']} B/lane  LDS {meta['group_segment_fixed_size']} B  "
                  f"waves/SIMD {co.waves_per_simd(meta['vgpr_count'])}")
            print(f"loop: {rep['instructions']} instructions, fences {rep['fences']}, vm waits {len(rep['vm_waits'])}, scratch in loop {len(rep['scratch_in_loop'])}, "
                  f"loads {rep['loads']}, stores {rep['stores']}")
            print(f"outer loop (static): {len(outer)} instructions: {mix}")


if __name__ == "__main__":
    main()
