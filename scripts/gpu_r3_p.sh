#!/bin/bash
mkdir -p gpurun_out/p
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_stream_persist_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/p/t_persist.log
timeout 900 python -m pytest tests/test_baseline_size_gpu.py -x -q -k streamer 2>&1 | tail -3 >> gpurun_out/p/t_persist.log
for rep in 1 2; do
  timeout 600 python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('persist', d['ms_per_step'], d['value'])
" >> gpurun_out/p/ab.log
done
timeout 300 python scripts/stream_step_trace.py 2>&1 | tail -36 > gpurun_out/p/trace.log
cat gpurun_out/p/*.log
