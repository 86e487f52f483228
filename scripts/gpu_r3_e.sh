#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -k "x6" > $O/e_test_gemm.log 2>&1; echo "test_gemm x6 rc=$?" | tee $O/e_summary.txt
tail -6 $O/e_test_gemm.log
timeout 900 python scripts/bench_gemm_x6.py $O/x6_sweep_e.jsonl > $O/e_x6_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/e_summary.txt
grep -E "^M=(768|1536)" $O/e_x6_sweep.log
for t in "768 1536 512 11" "768 1536 512 1" "768 512 1536 12 2" "768 512 1536 2 2"; do
  timeout 120 python scripts/gemm_x6_timeline.py $t 2>&1 | grep -v amdgpu.ids
done
