#!/bin/bash
mkdir -p gpurun_out/n
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_stream_persist_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/n/t_persist.log
timeout 1500 python -m pytest tests/test_streamer_gpu.py tests/test_baseline_size_gpu.py tests/test_properties_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/n/t_stream.log
cat gpurun_out/n/*.log
