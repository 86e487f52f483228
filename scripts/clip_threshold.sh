#!/bin/bash
# where the clip-per-XCD sampler starts to win: sampler time by batch size, launch path vs clip kernel forced from 2 clips on
out=gpurun_out/clip_threshold; mkdir -p $out; : > $out/times.log
for b in 2 3 4 5 6 7 9 10 12 13; do
  AFTER_SAMPLE_CLIP=0 timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/launch /" >> $out/times.log
  AFTER_SAMPLE_CLIP_MINB=2 timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/clip   /" >> $out/times.log
done
cat $out/times.log
