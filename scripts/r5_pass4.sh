#!/bin/bash
# gpurun -- bash scripts/r5_pass4.sh : width- and length-generic sample_seg_kernel (tiny: E = 256; nseg < 8 segments)
set -u
out=gpurun_out/r5_pass4; mkdir -p "$out"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sample_persist_gpu.py tests/test_denoiser_gpu.py tests/test_persist_protocol_gpu.py -x -q > "$out/test.log" 2>&1; tail -n 15 "$out/test.log"
for p in 0 1; do
  AFTER_SAMPLE_PERSIST=$p timeout 300 python scripts/time_sampler.py tiny 1 50 5 2>&1 | grep "sample " | sed "s/^/tiny persist=$p: /" | cut -c1-100
  AFTER_SAMPLE_PERSIST=$p timeout 300 python scripts/time_sampler.py base 1 50 5 2>&1 | grep "sample " | sed "s/^/base persist=$p: /" | cut -c1-100
done
timeout 600 python bench.py --from-audio --config tiny --steps 5 --warmup 2 --no-cpu-baseline > "$out/tiny_from_audio.json" 2> "$out/tiny.err"
python - <<PY
import json; d=json.loads(open("$out/tiny_from_audio.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("tiny from audio", d["ms_per_step"], d["value"], d["config"].get("sampler_path"), r.get("frac"), r.get("phase_us"))
PY
