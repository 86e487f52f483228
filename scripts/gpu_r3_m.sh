#!/bin/bash
# persistent streaming step: same-box A/B of the L2-warming fractions (bench.py --stream, 24 chunks each)
mkdir -p gpurun_out/m
export TMPDIR=/tmp
for rep in 1 2; do
for wm in "0,0,0" "4,16,4" "8,16,8" "0,16,0" "0,16,8"; do
  AFTER_STEP_WARM=$wm timeout 600 python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('warm=$wm', d['ms_per_step'], d['value'])
" >> gpurun_out/m/ab.log
done
done
cat gpurun_out/m/ab.log
