#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gemm_gpu.py tests/test_denoiser_gpu.py tests/test_properties_gpu.py tests/test_baseline_size_gpu.py -q > $O/f_test.log 2>&1; echo "pytest rc=$?" | tee $O/f_summary.txt
tail -5 $O/f_test.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/f_bench_b1.json 2> $O/f_bench_b1.err; echo "bench b1 rc=$?" | tee -a $O/f_summary.txt
python -c "import json; d=json.load(open('$O/f_bench_b1.json')); r=d['roofline']; print('b1', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['avg_launch_us'], r.get('fp32_mfma_path',{}).get('avg_launch_us'))"
timeout 600 python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/f_bench_b8.json 2> $O/f_bench_b8.err
python -c "import json; d=json.load(open('$O/f_bench_b8.json')); r=d['roofline']; print('b8', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['avg_launch_us'])"
for b in 1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st$b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --batch-per-gpu $b --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/st$b.log 2>&1)
  f=$(find $O/st$b -name "*kernel_stats.csv" | head -1)
  head -41 "$f" | cut -c1-300 > $O/f_x6_b${b}_kernel_stats.csv
  rm -rf $O/st$b
done
head -8 $O/f_x6_b1_kernel_stats.csv | cut -c1-220
