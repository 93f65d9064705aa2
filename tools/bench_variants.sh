#!/bin/bash
# run bench.py once per library variant in build_variants/ (same process conditions, sequential)
for f in build_variants/lib_*.so; do
  echo "== $f"
  TNCO_HIP_LIB=$PWD/$f python bench.py --steps 4 --warmup 1 --cpu-sample 0 --pmc 0 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('value %.4g  ms/step %.2f  frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"
done
