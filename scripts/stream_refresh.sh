set -u
O=gpurun_out/stream_r4b; mkdir -p $O; export TMPDIR=/tmp
python bench.py --stream --steps 16 --warmup 4 > $O/r4_bench_stream.json 2> $O/err.log
for p in 1 0 1 0; do
  AFTER_STREAM_PERSIST=$p python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'stream 8x100 steps', 'AFTER_STREAM_PERSIST': $p, 'ms_per_chunk': d['ms_per_step'], 'xrt': d['value']}))" >> $O/r4_ab_stream_persist.jsonl
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/sp -- python $GRAFT_REPO_ROOT/bench.py --stream --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/sp.log 2>&1)
f=$(find $O/sp -name "*kernel_stats.csv" | head -1)
head -31 "$f" | cut -c1-260 > $O/r4_bench_stream_kernel_stats.csv
rm -rf $O/sp
cat $O/r4_ab_stream_persist.jsonl; head -6 $O/r4_bench_stream_kernel_stats.csv | cut -c1-150
