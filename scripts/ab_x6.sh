# EXPERIMENT: the big Linears through gemm_x6.hip (AFTER_GEMM_X6=1) against the fp32 MFMA path, B = 8 and B = 1
for c in 0 1 0 1; do
  AFTER_GEMM_X6=$c python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > /tmp/o.json 2>/dev/null
  echo "b8 x6=$c $(python -c "import json; d=json.load(open('/tmp/o.json')); print(d['ms_per_step'], d['value'])")"
done
for c in 0 1; do
  AFTER_GEMM_X6=$c AFTER_GEMM_X6_MINROWS=512 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > /tmp/o.json 2>/dev/null
  echo "b1 x6=$c $(python -c "import json; d=json.load(open('/tmp/o.json')); print(d['ms_per_step'], d['value'])")"
done
AFTER_GEMM_X6=1 timeout 600 python -m pytest tests/test_baseline_size_gpu.py tests/test_denoiser_gpu.py tests/test_large_sizes_gpu.py -q -x 2>&1 | tail -2
