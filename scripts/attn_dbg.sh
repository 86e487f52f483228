# attention kernel time at B=8 under the AFTER_ATTN_DBG masks (1 no RoPE, 2 no reduce, 4 no LN tail, 8 no K/V loads)
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 8 15; do
  AFTER_ATTN_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ad -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch-per-gpu 8 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/ad -name "*kernel_stats.csv" | head -1)
  echo "dbg=$d $(grep attn_block $f | python3 -c "import sys,csv; [print(r[1], r[3]) for r in csv.reader(sys.stdin)]")"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/ad
done
