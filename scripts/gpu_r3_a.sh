#!/bin/bash
# round-3 GPU pass A: gemm_x6 correctness, tile sweep, sampler parity on both paths, first A/B bench
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q > $O/a_test_gemm.log 2>&1; echo "test_gemm rc=$?" | tee -a $O/a_summary.txt
tail -5 $O/a_test_gemm.log
timeout 900 python scripts/bench_gemm_x6.py $O/x6_sweep.jsonl > $O/a_x6_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/a_summary.txt
cat $O/a_x6_sweep.log
for t in "768 1536 512 1" "768 512 1536 2 2" "6144 1536 512 3" "6144 1536 512 5" "6144 512 1536 4 2"; do
  timeout 120 python scripts/gemm_x6_timeline.py $t >> $O/a_timeline.log 2>&1
done
cat $O/a_timeline.log
timeout 900 python -m pytest tests/test_denoiser_gpu.py tests/test_properties_gpu.py -x -q > $O/a_test_denoiser.log 2>&1; echo "test_denoiser rc=$?" | tee -a $O/a_summary.txt
tail -8 $O/a_test_denoiser.log
for c in 0 1; do
  AFTER_GEMM_X6=$c timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/a_b1_x6_$c.json 2>$O/a_b1_x6_$c.err
  echo "b1 x6=$c $(python -c "import json; d=json.load(open('$O/a_b1_x6_$c.json')); print(d['ms_per_step'], d['value'], d['roofline'])")" | tee -a $O/a_summary.txt
  AFTER_GEMM_X6=$c timeout 300 python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/a_b8_x6_$c.json 2>$O/a_b8_x6_$c.err
  echo "b8 x6=$c $(python -c "import json; d=json.load(open('$O/a_b8_x6_$c.json')); print(d['ms_per_step'], d['value'], d['roofline'])")" | tee -a $O/a_summary.txt
done
