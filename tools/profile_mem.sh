#!/bin/bash
# tools/profile_mem.sh TAG -- memory-request counters of the sweep kernel (one rocprofv3 --pmc run per
# group): L1->L2 requests, L2 hits/misses, L2->fabric (EA) read/write requests and their sizes, HBM
# bytes.  The kernel is bound by the NUMBER of random requests the memory system retires
# (tools/hbm_random.hip), so requests per move is the figure to watch.
#   PROF_SCRIPT=tools/time_fw.py PROF_KERNEL=fw_ tools/profile_mem.sh fw --replicas 65536 --sweeps 20
# profiles another script of this repository and lists the kernels whose name contains PROF_KERNEL.
set -u
TAG=${1:-mem}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SCRIPT=${PROF_SCRIPT:-bench.py}
KERNEL=${PROF_KERNEL:-sa_run_kernel}
if [ "$SCRIPT" = bench.py ]; then ARGS="--steps 2 --warmup 1 --cpu-sample 0 $*"; else ARGS="$*"; fi
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" FETCH_SIZE WRITE_SIZE "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -o pmc -- python3 "$ROOT/$SCRIPT" $ARGS > "$OUT/pmc_$N.log" 2>&1
done
python3 - "$OUT" "$KERNEL" <<'EOF'
import csv, glob, sys
from collections import defaultdict
pmc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if sys.argv[2] in name:
            pmc[(name.split("<")[0].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(pmc):
    v = pmc[k]
    print(f"{k[0]:28s} {k[1]:28s} {sum(v)/len(v):14.6g}  ({len(v)} launches)")
EOF
