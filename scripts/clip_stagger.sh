#!/bin/bash
# experiment: start offsets between the XCDs of the clip-per-XCD sampler (AFTER_CLIP_STAGGER, 10-ns ticks)
out=gpurun_out/clip_stagger; mkdir -p $out; : > $out/times.log
for s in 0 300 600 1200 2500 0; do
  AFTER_CLIP_STAGGER=$s timeout 300 python scripts/time_sampler.py base 8 50 5 2>&1 | grep "sample " | cut -c1-60 | sed "s/^/stagger $s: /" >> $out/times.log
done
cat $out/times.log
AFTER_CLIP_STAGGER=600 timeout 300 python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>&1 | grep -E "L5|step|qkv phase" | cut -c1-60
