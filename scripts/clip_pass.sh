#!/bin/bash
# The inner loop of a change to the clip-per-XCD offline sampler (sample_clip_kernel): its tests, same-box A/B against the
# launch path at 8 clips, per-phase trace.  gpurun -- bash scripts/clip_pass.sh [tests|time|all]
set -u
what=${1:-all}
out=gpurun_out/clip_pass
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$what" = tests ] || [ "$what" = all ]; then
    timeout 1500 python -m pytest tests/test_sample_clip_gpu.py -x -q > "$out/test.log" 2>&1
    tail -n 15 "$out/test.log"
fi
if [ "$what" = time ] || [ "$what" = all ]; then
    : > "$out/times.log"
    for rep in 1 2; do
        AFTER_SAMPLE_CLIP=0 timeout 300 python scripts/time_sampler.py base 8 50 5 2>&1 | grep "sample " | sed "s/^/launch: /" >> "$out/times.log"
        AFTER_SAMPLE_CLIP=1 timeout 300 python scripts/time_sampler.py base 8 50 5 2>&1 | grep "sample " | sed "s/^/clip:   /" >> "$out/times.log"
    done
    cut -c1-100 "$out/times.log"
    timeout 300 python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 --detail > "$out/trace_xcd3.txt" 2>&1
    grep -v amdgpu.ids "$out/trace_xcd3.txt" | cut -c1-150 | tail -n 40
fi
