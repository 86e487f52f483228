#!/bin/bash
# gpurun -- bash scripts/r5_pass11.sh : tiny width on the one-clip persistent sampler with the weight tiles warmed into the L2
set -u
out=gpurun_out/r5_pass11; mkdir -p "$out"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sample_persist_gpu.py -x -q -k "tiny or geometries" > "$out/test.log" 2>&1; tail -n 4 "$out/test.log"
for r in 1 2; do
timeout 300 python scripts/time_sampler.py tiny 1 50 7 2>&1 | grep "sample " | cut -c1-80
AFTER_T=128 timeout 300 python scripts/time_sampler.py tiny 1 50 7 2>&1 | grep "sample " | sed "s/^/T=128 /" | cut -c1-80
done
timeout 300 python scripts/time_sampler.py base 1 50 7 2>&1 | grep "sample " | cut -c1-80
