"""What do rocprofv3's memory-side counters report for RANDOM 32- / 64- / 128-byte accesses on gfx950?

    python tools/pmc_calibrate.py [--out profiles/r04_pmc_calibration.md]

Runs build_variants/pmc_calibrate (tools/pmc_calibrate.hip: kernels that move a known number of bytes in the access
patterns of the sweep kernels) under `rocprofv3 --pmc`, one pass per counter group, and tabulates per kernel and
counter: the raw value, and -- for the byte counters -- bytes counted / bytes moved.  The table answers whether
FETCH_SIZE must be doubled for these patterns (the guide's rule for 128-byte streaming requests) and what one
TCC_EA0_RDREQ / WRREQ stands for.  bench.py's `roofline.traffic` uses the resulting rule (bench.traffic_fields).
"""
import argparse
import csv
import glob
import os
import pathlib
import shutil
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parents[1]
GROUPS = (("FETCH_SIZE",), ("WRITE_SIZE",), ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"),
          ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"), ("TCC_REQ_sum", "TCC_MISS_sum"),
          ("TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"), ("TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum"))
ACC = 2048 * 64 * 256
MOVED = {"cal_random_read<32>": ("read", ACC * 32), "cal_random_read<64>": ("read", ACC * 64),
         "cal_random_read<128>": ("read", ACC * 128), "cal_random_write<4>": ("write", ACC * 4),
         "cal_random_write<32>": ("write", ACC * 32), "cal_random_write<64>": ("write", ACC * 64),
         "cal_random_write<128>": ("write", ACC * 128), "cal_stream_read": ("read", (2 << 26) * 16),
         "cal_stream_write": ("write", (2 << 26) * 16)}


def short(name):
    for k in MOVED:
        if k.split("<")[0] in name and (("<" not in k) or (k[k.index("<"):] in name.replace("<(int)", "<").replace("<int=", "<"))):
            return k
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    exe = ROOT / "build_variants" / "pmc_calibrate"
    if not exe.exists():
        sys.exit(f"{exe} is missing: hipcc -O3 --offload-arch=gfx950 -o {exe} tools/pmc_calibrate.hip")
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="tnco_cal_", dir="/tmp"))
    vals, notes = {}, []
    for gi, grp in enumerate(GROUPS):
        out = tmp / f"g{gi}"
        cmd = [prof, "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", str(out), "-o", "cal", "--", str(exe)]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            notes.append(f"{grp}: timed out")
            continue
        files = glob.glob(str(out / "**" / "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            notes.append(f"{grp}: rocprofv3 rc {p.returncode}: {p.stderr.strip().splitlines()[-1] if p.stderr.strip() else ''}")
            continue
        for f in files:
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    if k:
                        vals[(k, r["Counter_Name"])] = vals.get((k, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
    shutil.rmtree(tmp, ignore_errors=True)
    ctrs = [c for g in GROUPS for c in g if any((k, c) in vals for k in MOVED)]
    lines = ["# rocprofv3 memory-side counters against known byte counts, random 32 / 64 / 128-byte pieces (gfx950)", "",
             f"`tools/pmc_calibrate.py` -> `build_variants/pmc_calibrate` (tools/pmc_calibrate.hip): {ACC} accesses per random kernel "
             "by groups of 4 lanes into an 8-GiB buffer; 2 GiB streamed at 16 bytes per lane.  FETCH_SIZE / WRITE_SIZE are in KiB.", "",
             "| kernel | bytes moved | " + " | ".join(ctrs) + " | FETCH_SIZE bytes / moved | WRITE_SIZE bytes / moved | RDREQ / access | WRREQ / access |",
             "|---|---|" + "---|" * (len(ctrs) + 4)]
    for k, (kind, moved) in MOVED.items():
        row = [k, f"{moved:.4g}"] + [f"{vals[(k, c)]:.5g}" if (k, c) in vals else "" for c in ctrs]
        f_ = vals.get((k, "FETCH_SIZE"))
        w_ = vals.get((k, "WRITE_SIZE"))
        n_acc = ACC if "random" in k else moved / 64
        row.append(f"{f_ * 1024 / moved:.3f}" if f_ is not None and kind == "read" else "")
        row.append(f"{w_ * 1024 / moved:.3f}" if w_ is not None and kind == "write" else "")
        row.append(f"{vals[(k, 'TCC_EA0_RDREQ_sum')] / n_acc:.3f}" if (k, "TCC_EA0_RDREQ_sum") in vals else "")
        row.append(f"{vals[(k, 'TCC_EA0_WRREQ_sum')] / n_acc:.3f}" if (k, "TCC_EA0_WRREQ_sum") in vals else "")
        lines.append("| " + " | ".join(row) + " |")
    if notes:
        lines += ["", "Counter groups that could not be collected: " + "; ".join(notes)]
    lines += ["", "(streaming rows: 'per access' = per 64 bytes moved)"]
    txt = "\n".join(lines) + "\n"
    print(txt)
    if a.out:
        pathlib.Path(a.out).write_text(txt)


if __name__ == "__main__":
    main()
