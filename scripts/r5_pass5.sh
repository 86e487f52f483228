#!/bin/bash
# gpurun -- bash scripts/r5_pass5.sh : clip sampler with the attention inside the qkv tiles (AFTER_CLIP_FUSE=0/1)
set -u
out=gpurun_out/r5_pass5; mkdir -p "$out"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sample_clip_gpu.py -x -q > "$out/test.log" 2>&1; tail -n 12 "$out/test.log"
for rep in 1 2; do
  for cfg in base midi; do
    AFTER_CLIP_FUSE=0 timeout 300 python scripts/time_sampler.py $cfg 8 50 3 2>&1 | grep "sample " | sed "s/^/items: /" | cut -c1-80
    timeout 300 python scripts/time_sampler.py $cfg 8 50 3 2>&1 | grep "sample " | sed "s/^/fused: /" | cut -c1-80
  done
done
timeout 300 python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 > "$out/clip_trace.txt" 2>&1; grep -v amdgpu.ids "$out/clip_trace.txt" | cut -c1-60 | sed -n 2,14p; tail -6 "$out/clip_trace.txt"
