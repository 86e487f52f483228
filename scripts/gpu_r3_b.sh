#!/bin/bash
# round-3 GPU pass B: rocprof kernel stats of the sampler on the default (bf16-split) path, B = 1 and 8
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
for b in 1 8; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st$b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --batch-per-gpu $b --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/st$b.log 2>&1)
  f=$(find $O/st$b -name "*kernel_stats.csv" | head -1)
  head -41 "$f" | cut -c1-300 > $O/b_x6_b${b}_kernel_stats.csv
  rm -rf $O/st$b
done
cat $O/b_x6_b1_kernel_stats.csv | cut -c1-200 | head -24
cat $O/b_x6_b8_kernel_stats.csv | cut -c1-200 | head -16
