#!/bin/bash
# gpurun -- bash scripts/r5_pass6.sh : the fused clip sampler under the baseline-size / protocol tests, bench legs at 8 clips
set -u
out=gpurun_out/r5_pass6; mkdir -p "$out"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_baseline_size_gpu.py tests/test_persist_protocol_gpu.py tests/test_sample_clip_gpu.py tests/test_bench_multirank_gpu.py -x -q > "$out/test.log" 2>&1; tail -n 8 "$out/test.log"
timeout 600 python bench.py --batch-per-gpu 8 --steps 6 --warmup 2 --no-cpu-baseline > "$out/b8.json" 2> "$out/b8.err"
timeout 600 python bench.py --config midi --batch-per-gpu 8 --steps 4 --warmup 1 --no-cpu-baseline > "$out/midi_b8.json" 2> "$out/midi_b8.err"
python - <<PY
import json
for f in ("b8", "midi_b8"):
    d=json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["ms_per_step"], d["value"], r["frac"], r.get("phase_us"), r.get("avg_launch_us"), r.get("gemm_phases", {}).get("frac"))
PY
