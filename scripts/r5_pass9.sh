#!/bin/bash
# gpurun -- bash scripts/r5_pass9.sh : GroupNorm -> Snake -> k = 1 conv as one launch (conv1_act_kernel): parity, same-box A/B
set -u
out=gpurun_out/r5_pass9; mkdir -p "$out"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_autoencoder_gpu.py tests/test_baseline_size_gpu.py tests/test_conv_tm_gpu.py tests/test_encoder_stream_gpu.py -x -q > "$out/test.log" 2>&1; tail -n 6 "$out/test.log"
for m in 0 1 0 1; do
  AFTER_AE_FUSE_K1=$m timeout 300 python scripts/time_codec.py --rounds 20 2>/dev/null | grep workload | python -c "import json,sys
for l in sys.stdin:
    d=json.loads(l); d['AFTER_AE_FUSE_K1']=$m; print(json.dumps(d))" >> "$out/r5_ab_conv1_act.jsonl"
done
cat "$out/r5_ab_conv1_act.jsonl" | cut -c1-260
