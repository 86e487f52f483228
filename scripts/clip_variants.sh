#!/bin/bash
# per-phase traces of the clip-per-XCD sampler under the kernel variants built by scripts/build_variant.sh (same box)
out=gpurun_out/clip_variants; mkdir -p $out
cp after_amd/lib/libafter_hip.so $out/default.so
for v in default $(ls scripts/variants); do
  if [ "$v" = default ]; then cp $out/default.so after_amd/lib/libafter_hip.so; else cp scripts/variants/$v/libafter_hip.so after_amd/lib/libafter_hip.so; fi
  timeout 300 python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 > $out/trace_$v.txt 2>&1
  echo "== $v"; grep -E "L5|XCD 3 step|clock" $out/trace_$v.txt | cut -c1-48
done
cp $out/default.so after_amd/lib/libafter_hip.so; rm $out/default.so
