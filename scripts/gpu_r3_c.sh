#!/bin/bash
# round-3 GPU pass C: full -m gpu suite + smoke + bench legs on the per-Linear dispatch
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/c_test_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/c_summary.txt
tail -15 $O/c_test_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee -a $O/c_summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/c_bench_b1.json 2> $O/c_bench_b1.err; echo "bench b1 rc=$?" | tee -a $O/c_summary.txt
cat $O/c_bench_b1.json; tail -3 $O/c_bench_b1.err
timeout 600 python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/c_bench_b8.json 2> $O/c_bench_b8.err; echo "bench b8 rc=$?" | tee -a $O/c_summary.txt
cat $O/c_bench_b8.json; tail -3 $O/c_bench_b8.err
timeout 600 python bench.py --steps 5 --warmup 2 --from-audio --config tiny --no-cpu-baseline > $O/c_bench_tiny_audio.json 2> $O/c_bench_tiny_audio.err; echo "bench from-audio rc=$?" | tee -a $O/c_summary.txt
cat $O/c_bench_tiny_audio.json; tail -3 $O/c_bench_tiny_audio.err
