#!/bin/bash
# gpurun -- bash scripts/r5_pass10.sh : per-launch durations of the fused k = 1 convs against act_pad + conv
set -u
out=gpurun_out/r5_pass10; mkdir -p "$out"; export TMPDIR=/tmp
for m in 0 1; do
  for b in 1 8; do
  (cd /tmp && AFTER_AE_FUSE_K1=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/tr$m$b -- python $GRAFT_REPO_ROOT/scripts/time_codec.py --rounds 3 --batches $b --only decode > $GRAFT_REPO_ROOT/$out/tr.log 2>&1)
  f=$(find $out/tr$m$b -name "*kernel_stats.csv" | head -1)
  echo "== FUSE_K1=$m B=$b"
  python - "$f" <<'PY'
import csv,sys
for i,r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i>=16: break
    n=r["Name"].replace("after::(anonymous namespace)::","")
    print(f'{n[:70]:70s} calls {r["Calls"]:>5s} total_us {float(r["TotalDurationNs"])/1e3:9.1f} avg_us {float(r["AverageNs"])/1e3:7.1f}')
PY
  rm -rf $out/tr$m$b
  done
done
