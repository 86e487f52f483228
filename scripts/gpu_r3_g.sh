#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gemm_gpu.py -q -k "x6" > $O/g_test_gemm.log 2>&1; echo "test_gemm x6 rc=$?" | tee $O/g_summary.txt
tail -4 $O/g_test_gemm.log
timeout 900 python scripts/bench_gemm_x6.py $O/x6_sweep_g.jsonl > $O/g_x6_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/g_summary.txt
grep -E "^M=(3072|6144|12288)" $O/g_x6_sweep.log
timeout 600 python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/g_bench_b8.json 2> $O/g_bench_b8.err
python -c "import json; d=json.load(open('$O/g_bench_b8.json')); r=d['roofline']; print('b8', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['avg_launch_us'])"
timeout 900 python -m pytest tests/test_baseline_size_gpu.py -q -k "b8 or B8 or shard or midi" > $O/g_test_b8.log 2>&1; echo "b8 tests rc=$?"; tail -3 $O/g_test_b8.log
