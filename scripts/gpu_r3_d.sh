#!/bin/bash
# round-3 GPU pass D: full -m gpu suite after the parity / determinism / bottleneck work + x6 error sweep
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $O/d_test_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/d_summary.txt
tail -40 $O/d_test_gpu.log
timeout 900 python scripts/bench_gemm_x6.py $O/x6_sweep.jsonl > $O/d_x6_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/d_summary.txt
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r3/x6_sweep.jsonl")]
kp={1:2,2:4,3:1,4:2,5:1,6:2,7:1,8:2,9:1}
for r in rows:
    s=r["fp32_err_by_kparts_max_rms"][str(kp[r["tile"]])] if isinstance(list(r["fp32_err_by_kparts_max_rms"].keys())[0],str) else r["fp32_err_by_kparts_max_rms"][kp[r["tile"]]]
    print(r["M"],r["N"],r["K"],r["epilogue"],"tile",r["tile"],"us",r["us"],"max ratio %.2f rms ratio %.2f | vs default max %.2f rms %.2f"%(r["err_vs_fp64"]/s[0], r["rms_err_vs_fp64"]/s[1], r["err_vs_fp64"]/r["fp32_kernel_err"], r["rms_err_vs_fp64"]/r["fp32_kernel_rms_err"]))
PY
