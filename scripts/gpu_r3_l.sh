#!/bin/bash
# persistent streaming step: parity, then the per-phase timeline for a few L2-warming settings, then the bench A/B
mkdir -p gpurun_out/l
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_baseline_size_gpu.py -x -q -k streamer 2>&1 | tail -3 > gpurun_out/l/t_cfg5.log
for wm in "0,0,0" "4,16,4" "8,16,8"; do
  echo "== warm $wm" >> gpurun_out/l/trace.log
  AFTER_STEP_WARM=$wm timeout 300 python scripts/stream_step_trace.py 2>&1 | tail -9 >> gpurun_out/l/trace.log
done
for p in 1 0; do
  AFTER_STREAM_PERSIST=$p timeout 600 python bench.py --stream --steps 16 --warmup 4 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('persist=$p', d['ms_per_step'], d['value'])
" >> gpurun_out/l/ab.log
done
cat gpurun_out/l/*.log
