# same-box A/B: the L2-warming touches as LDS-DMA loads into an LDS sink (fix) against VGPR-sink inline asm (before)
cp after_amd/lib/libafter_hip.so /tmp/fix.so
for rep in 1 2 3; do
  for v in before fix; do
    if [ $v = before ]; then cp scripts/variants/prewarmfix/libafter_hip.so after_amd/lib/libafter_hip.so; else cp /tmp/fix.so after_amd/lib/libafter_hip.so; fi
    for cfg in "base 1 7" "tiny 1 7" "base 2 5"; do set -- $cfg; python scripts/time_sampler.py $1 $2 50 $3 2>/dev/null | tail -1 | cut -c1-75 | sed "s/^/$v: /"; done
    python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v: stream', d['ms_per_step'], 'ms per chunk')"
  done
done
cp /tmp/fix.so after_amd/lib/libafter_hip.so
