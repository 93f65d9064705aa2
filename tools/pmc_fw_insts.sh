#!/bin/bash
# tools/pmc_fw_insts.sh [VARIANT...] -- instruction counters of the finite-width leg's kernels (rocprofv3 --pmc, one pass per
# counter pair), for the library in the tree and for build_variants/lib_VARIANT.so: per kernel and REPLICA-launch the
# VALU / SALU / LDS / VMEM / SMEM instructions a wavefront executes, and the busy fractions.  The early-exit variants
# (-DTNCO_FWW_STOP=k: fw_wave_kernel hands the replica to the fallback kernels after phase k) give the per-phase
# instruction counts of the re-slice by difference.  Output: gpurun_out/r06/pmc_fw_insts.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-4}; WARM=${WARM:-2}
for V in tree "$@"; do
  if [ "$V" = tree ]; then unset TNCO_HIP_LIB; else export TNCO_HIP_LIB=$ROOT/build_variants/lib_$V.so; fi
  rm -rf /tmp/pfi_*
  if [ "$V" = tree ]; then
    GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SMEM SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_WAIT_ANY")
  else  # (the early-exit variants: instruction counts only)
    GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES")
  fi
  for C in "${GROUPS_[@]}"; do
    N=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pfi_$N -o pmc -- python3 "$ROOT/bench.py" --workload fw --pmc 0 --cpu-sample 0 --e2e 0 --calibrate 0 --no-validate --steps $STEPS --warmup $WARM > /tmp/pfi_$N.log 2>&1 || echo "pass $C failed: $(tail -c 300 /tmp/pfi_$N.log)"
  done
  python3 - "$V" <<'PY' | tee -a "$OUT/pmc_fw_insts.txt"
import csv, glob, sys
from collections import defaultdict
pmc = defaultdict(list)
for f in glob.glob("/tmp/pfi_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tnco::", "")
        if any(k in name for k in ("fw_wave", "sa_run_kernel", "fw_reslice")):
            pmc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
kernels = sorted({k for k, _ in pmc})
print(f"== library: {sys.argv[1]}")
for k in kernels:
    c = {ctr: v for (kk, ctr), v in pmc.items() if kk == k}
    n = len(next(iter(c.values())))
    waves = sum(c.get("SQ_WAVES", [0])) / max(n, 1)
    per = lambda ctr: sum(c.get(ctr, [0])) / max(sum(c.get("SQ_WAVES", [1])), 1)
    print(f"{k}: {n} dispatches, {waves:.0f} wavefronts each; per wavefront: VALU {per('SQ_INSTS_VALU'):.0f}  SALU {per('SQ_INSTS_SALU'):.0f}  "
          f"LDS {per('SQ_INSTS_LDS'):.0f}  SMEM {per('SQ_INSTS_SMEM'):.0f}  VMEM rd {per('SQ_INSTS_VMEM_RD'):.0f} wr {per('SQ_INSTS_VMEM_WR'):.0f}; "
          f"wave cycles {per('SQ_WAVE_CYCLES'):.0f}, waiting {per('SQ_WAIT_ANY'):.0f} (LDS {per('SQ_WAIT_INST_LDS'):.0f}); "
          f"VALU busy {sum(c.get('SQ_ACTIVE_INST_VALU', [0])) / max(sum(c.get('SQ_BUSY_CYCLES', [1])), 1):.3f}, "
          f"scalar busy {sum(c.get('SQ_ACTIVE_INST_SCA', [0])) / max(sum(c.get('SQ_BUSY_CYCLES', [1])), 1):.3f}")
PY
done
