#!/bin/bash
# gpurun -- bash scripts/r5_pass3.sh : state of HEAD at 8 clips (bench legs, per-phase trace of the clip sampler), tiny, stream
set -u
out=gpurun_out/r5_pass3; mkdir -p "$out"; export TMPDIR=/tmp
timeout 600 python bench.py --batch-per-gpu 8 --steps 6 --warmup 2 --no-cpu-baseline > "$out/b8.json" 2> "$out/b8.err"
python - <<PY
import json; d=json.loads(open("$out/b8.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("b8", d["ms_per_step"], d["value"], r["frac"], r.get("phase_us"), r.get("avg_launch_us"), d["codec"] if "codec" in d else r.get("codec"))
PY
timeout 600 python bench.py --config midi --batch-per-gpu 8 --steps 4 --warmup 1 --no-cpu-baseline > "$out/midi_b8.json" 2> "$out/midi_b8.err"
python - <<PY
import json; d=json.loads(open("$out/midi_b8.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("midi b8", d["ms_per_step"], d["value"], r["frac"], r.get("phase_us"), r.get("avg_launch_us"))
PY
for cfg in base midi; do
  timeout 300 python scripts/time_sampler.py $cfg 8 50 3 2>&1 | grep "sample " | sed "s/^/clip $cfg: /" | cut -c1-100
done
timeout 300 python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 > "$out/clip_trace.txt" 2>&1; grep -v amdgpu.ids "$out/clip_trace.txt" | cut -c1-60 | head -20; tail -6 "$out/clip_trace.txt"
timeout 300 python scripts/time_codec.py --rounds 20 2>/dev/null | grep workload | cut -c1-220
