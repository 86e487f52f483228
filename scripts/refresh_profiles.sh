#!/bin/bash
# After a change to the sampler's kernel sources: the artefacts whose validity depends on the source hash (PMC traffic, the
# bench lines that quote it) and the streaming sampler's, on one lease.  bash scripts/refresh_profiles.sh r3
R=${1:-r3}
O=gpurun_out/refresh_$R
mkdir -p $O
export TMPDIR=/tmp
python bench.py --pmc > $O/pmc_b1.log 2>&1
python bench.py --pmc --batch-per-gpu 8 > $O/pmc_b8.log 2>&1
cp profiles/${R}_pmc_hbm_base_b1.json profiles/${R}_pmc_hbm_base_b8.json $O/
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_b1.json 2> $O/bench_b1.err
python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/${R}_bench_b8.json 2> $O/bench_b8.err
python bench.py --steps 5 --warmup 2 --from-audio --no-cpu-baseline > $O/${R}_bench_base_from_audio.json 2> $O/bench_fa.err
python bench.py --stream --steps 16 --warmup 4 > $O/${R}_bench_stream.json 2> $O/bench_stream.err
for p in 1 0 1 0; do
  AFTER_STREAM_PERSIST=$p python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'stream 8x100 steps', 'AFTER_STREAM_PERSIST': $p, 'ms_per_chunk': d['ms_per_step'], 'xrt': d['value']}))" >> $O/${R}_ab_stream_persist.jsonl
done
python scripts/stream_step_trace.py > $O/${R}_stream_step_trace.txt 2>/dev/null
python scripts/stream_step_trace.py --offline > $O/${R}_offline_step_trace.txt 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/sp -- python $GRAFT_REPO_ROOT/bench.py --stream --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/sp.log 2>&1)
f=$(find $O/sp -name "*kernel_stats.csv" | head -1)
head -31 "$f" | cut -c1-260 > $O/${R}_bench_stream_kernel_stats.csv
rm -rf $O/sp
ls $O
