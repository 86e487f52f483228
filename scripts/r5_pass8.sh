#!/bin/bash
# gpurun -- bash scripts/r5_pass8.sh : the opt-in bf16 tolerance tier (tests, sampler times against the default arithmetic)
set -u
out=gpurun_out/r5_pass8; mkdir -p "$out"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_tier_gpu.py -x -q > "$out/test.log" 2>&1; tail -n 12 "$out/test.log"
for b in 1 8; do
  for m in 1 3; do
    AFTER_TIME_GEMM_PATH=$m timeout 300 python scripts/time_sampler.py base $b 50 5 2>&1 | grep "sample " | sed "s/^/gemm path $m: /" | cut -c1-90
  done
done
