#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for p in 1 0; do
  AFTER_GEMM_X6_PERSIST=$p timeout 600 python bench.py --steps 8 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/h_b8_p$p.json 2> $O/h_b8.err
  python -c "import json; d=json.load(open('$O/h_b8_p$p.json')); r=d['roofline']; print('b8 persist=$p', d['ms_per_step'], d['value'], r['avg_launch_us'])"
done
done
