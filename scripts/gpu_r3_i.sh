#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_denoiser_gpu.py -q > $O/i_test.log 2>&1; echo "pytest rc=$?" | tee $O/i_summary.txt
tail -3 $O/i_test.log
for t in "768 1536 512 11" "768 1536 512 11 1" "768 512 1536 12 2"; do
  timeout 120 python scripts/gemm_x6_timeline.py $t 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/i_bench_b1.json 2> $O/i_bench_b1.err
python -c "import json; d=json.load(open('$O/i_bench_b1.json')); r=d['roofline']; print('b1', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['avg_launch_us'])"
done
timeout 600 python bench.py --steps 8 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/i_bench_b8.json 2> $O/i_bench_b8.err
python -c "import json; d=json.load(open('$O/i_bench_b8.json')); r=d['roofline']; print('b8', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['avg_launch_us'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st1 -- python $GRAFT_REPO_ROOT/scripts/time_sampler.py > $GRAFT_REPO_ROOT/$O/st1.log 2>&1)
f=$(find $O/st1 -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; rm -rf $O/st1
