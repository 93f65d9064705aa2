#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<TAG>/ (tools/profile.sh) into profiles/<TAG>_summary.{md,json}.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are
collected in separate passes and reported in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced
streams by 2x (64 B counted per 128-B request).  This kernel's reads are mostly <= 128-B rows and
32-B records rather than 16 B/lane streams, so both the raw and the doubled figure are reported.
"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / f"prof_{tag}"
out = {}
lines = [f"# rocprofv3 summary {tag}", ""]

def find(pattern):
    return sorted(glob.glob(str(src / pattern), recursive=True))

# kernel stats
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    lines += ["## kernel stats (rocprofv3 --kernel-trace --stats)", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    ks = []
    for r in rows:
        name = r.get("Name", "")
        short = name.split("(")[0][-90:]
        calls = int(r.get("Calls", 0))
        tot = float(r.get("TotalDurationNs", 0)) / 1e6
        avg = float(r.get("AverageNs", 0)) / 1e6
        pct = r.get("Percentage", "")
        lines.append(f"| `{short}` | {calls} | {tot:.3f} | {avg:.3f} | {pct} |")
        ks.append(dict(name=name, calls=calls, total_ms=tot, avg_ms=avg))
    out["kernel_stats"] = ks
    lines.append("")

# per-dispatch registers from the trace
for f in find("trace/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "sa_run_kernel" in r.get("Kernel_Name", "")]
    if rows:
        r = rows[-1]
        out["sa_run_dispatch"] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        lines += ["## sa_run_kernel dispatch", "", "```", json.dumps(out["sa_run_dispatch"]), "```", ""]

# counters
pmc = defaultdict(list)
for f in find("pmc_*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sa_run_kernel" in r.get("Kernel_Name", ""):
            pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if pmc:
    lines += ["## PMC counters, sa_run_kernel, mean per launch", "", "| counter | mean per launch | launches |", "|---|---|---|"]
    means = {}
    for k in sorted(pmc):
        v = pmc[k]
        # rocprofv3 emits one row per dispatch per counter (summed over XCDs/dims)
        means[k] = sum(v) / len(v)
        lines.append(f"| {k} | {means[k]:.6g} | {len(v)} |")
    out["pmc_mean_per_launch"] = means
    lines.append("")
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        rd, wr = means["FETCH_SIZE"] * 1024, means["WRITE_SIZE"] * 1024
        out["hbm_bytes_per_launch_raw"] = rd + wr
        out["hbm_bytes_per_launch"] = 2 * rd + wr
        lines += [f"HBM bytes per launch: read {rd:.4g} B raw (x2 gfx950 correction = {2*rd:.4g}), write {wr:.4g} B; "
                  f"corrected total {2*rd+wr:.4g} B", ""]
(root / "profiles").mkdir(exist_ok=True)
(root / "profiles" / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
(root / "profiles" / f"{tag}_summary.json").write_text(json.dumps(out, indent=1) + "\n")
print("\n".join(lines))
