"""Register / LDS budget of the kernels whose occupancy the measured numbers depend on (cross-compiles; no GPU needed).

    python tools/check_registers.py            # ~4 minutes: two translation units, device code only

fw_wave_kernel sits at 127 VGPRs -- one more wavefront per SIMD than at 129 (round 4 lost 20 % of the re-slice that way
twice before noticing); sa_run_kernel at 168 (three wavefronts per SIMD).  Exits non-zero when a kernel leaves its budget.
"""
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
CSRC = ROOT / "tnco_amd" / "csrc"
# mangled-name fragment -> (what, max VGPRs, min waves per SIMD)
BUDGET = {
    ("inst_2_3.hip", "sa_run_kernelILi2ELi3ELb0ELb0ELb0E"): ("sa_run_kernel<2, 3, false, false, false> (headline leg)", 168, 3),
    ("inst_2_4.hip", "sa_run_fw_kernelILi2ELi4ELb0E"): ("sa_run_fw_kernel<2, 4, false> (finite-width moves)", 192, 2),
    ("inst_2_4.hip", "fw_wave_kernelILi9ELi4ELb0ELb0E"): ("fw_wave_kernel<9, 4, false, false> (config 5 re-slice)", 128, 4),
    ("inst_2_4.hip", "fw_wave_kernelILi6ELi4ELb1ELb0E"): ("fw_wave_kernel<6, 4, true, false> (hyper-index networks up to 384 nodes)", 128, 4),
}


def main():
    bad = 0
    for tu in sorted({k[0] for k in BUDGET}):
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
               "--cuda-device-only", "-c", "-o", "/dev/null", tu, "-Rpass-analysis=kernel-resource-usage"]
        txt = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
        blocks = txt.split("remark: Function Name: ")
        for (t, frag), (what, max_v, min_w) in BUDGET.items():
            if t != tu:
                continue
            blk = next((b for b in blocks if frag in b.splitlines()[0]), None)
            if blk is None:
                print(f"{what}: not found in {tu}")
                bad += 1
                continue
            get = lambda key: int(re.search(key + r": (\d+)", blk).group(1))  # noqa: E731
            v, w, sc, sp = get("VGPRs"), get(r"Occupancy \[waves/SIMD\]"), get(r"ScratchSize \[bytes/lane\]"), get("VGPRs Spill")
            ok = v <= max_v and w >= min_w
            bad += not ok
            print(f"{'ok  ' if ok else 'OVER'} {what}: {v} VGPRs (budget {max_v}), {w} wavefronts per SIMD (at least {min_w}), "
                  f"scratch {sc} B/lane, {sp} VGPRs spilled")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
