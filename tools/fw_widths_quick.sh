#!/bin/bash
# the finite-width leg of bench.py at max_width 28 / 32 / 40 (+ the easier network of rounds 1-3), as tools/r04_measure.sh
# runs it (25 steps of 100 sweeps, the first 5 untimed); TNCO_HIP_LIB=... for another build of the library
for cfg in "supremacy 28" "supremacy 32" "supremacy 40" "alternating 40"; do
  set -- $cfg
  timeout 300 python bench.py --workload fw --fw-layout $1 --fw-max-width $2 --pmc 0 --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1 $2', round(j['value']/1e9,3), 'e9 move-evals/s', round(j['ms_per_step'],2), 'ms/step  fell back', r['reslices']['fell_back'], 'of', r['reslices']['repriced'])"
done
