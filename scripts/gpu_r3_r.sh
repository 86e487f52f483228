#!/bin/bash
mkdir -p gpurun_out/r
export TMPDIR=/tmp
for p in 1 0 1 0; do
  AFTER_SAMPLE_PERSIST=$p timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('sample_persist=$p', d['ms_per_step'], d['value'])
" >> gpurun_out/r/ab.log
done
cat gpurun_out/r/ab.log
