cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pg_$N -o pmc -- python3 $ROOT/tools/${GREEDY_SCRIPT:-time_greedy.py} ${GREEDY_ARGS-512 65536} > /tmp/pg_$N.log 2>&1
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
pmc = defaultdict(list)
for f in glob.glob("/tmp/pg_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "greedy" in name or "shuffle" in name:
            pmc[(name.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(pmc):
    v = pmc[k]
    print(k, len(v), max(v))
PY
