#!/bin/bash
# persistent streaming step: parity on both paths, then same-box A/B of the stream bench
mkdir -p gpurun_out/k
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_streamer_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/k/t_streamer.log
timeout 900 python -m pytest tests/test_baseline_size_gpu.py -x -q -k streamer 2>&1 | tail -5 > gpurun_out/k/t_cfg5.log
AFTER_STREAM_PERSIST=0 timeout 900 python -m pytest tests/test_streamer_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/k/t_streamer_launch.log
for p in 1 0 1 0; do
  AFTER_STREAM_PERSIST=$p timeout 600 python bench.py --stream --steps 16 --warmup 4 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('persist=$p', d['ms_per_step'], d['value'], d.get('config'))
" >> gpurun_out/k/ab.log
done
cat gpurun_out/k/*.log
