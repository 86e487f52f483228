#!/bin/bash
# gpurun -- bash scripts/r5_pass2.sh : pair attention over two key tiles (midi) + the looping act_pad_x6 (codec at 8 clips)
set -u
out=gpurun_out/r5_pass2; mkdir -p "$out"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sample_clip_gpu.py tests/test_autoencoder_gpu.py tests/test_conv_tm_gpu.py -x -q > "$out/test.log" 2>&1; tail -n 5 "$out/test.log"
for rep in 1 2; do
  for cfg in base midi; do
    AFTER_SAMPLE_CLIP=0 timeout 300 python scripts/time_sampler.py $cfg 8 50 3 2>&1 | grep "sample " | sed "s/^/launch: /" | cut -c1-80
    timeout 300 python scripts/time_sampler.py $cfg 8 50 3 2>&1 | grep "sample " | sed "s/^/clip:   /" | cut -c1-80
  done
done
for m in 0 16 0 16 8 32; do
  echo "AFTER_ACT_LOOP=$m"; AFTER_ACT_LOOP=$m timeout 300 python scripts/time_codec.py --rounds 20 2>/dev/null | grep workload | cut -c1-200
done
