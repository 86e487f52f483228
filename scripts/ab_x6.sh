# the qkv / MLP Linears on gemm_x6.hip (default) against the fp32 MFMA kernel (AFTER_GEMM_X6=0), B = 8 and B = 1, same box
for c in 0 1 0 1; do
  AFTER_GEMM_X6=$c python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > /tmp/o.json 2>/dev/null
  echo "b8 x6=$c $(python -c "import json; d=json.load(open('/tmp/o.json')); print(d['ms_per_step'], d['value'])")"
done
for c in 0 1 0 1; do
  AFTER_GEMM_X6=$c python bench.py --steps 12 --warmup 3 --no-cpu-baseline > /tmp/o.json 2>/dev/null
  echo "b1 x6=$c $(python -c "import json; d=json.load(open('/tmp/o.json')); print(d['ms_per_step'], d['value'])")"
done
