#!/bin/bash
# Regenerates the per-round measurement artefacts on the GPU box (run through gpurun from the repo root):
#   bash scripts/round_profiles.sh r2
# Writes into gpurun_out/final_<round>/; the summaries are then copied into profiles/ by hand.
R=${1:-r2}
O=gpurun_out/final_$R
mkdir -p $O
export TMPDIR=/tmp
python bench.py --pmc > $O/pmc.log 2>&1   # first: the bench line below quotes its traffic figure
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_b1.json 2> $O/bench_b1.err
python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/${R}_bench_b8.json 2> $O/bench_b8.err
python bench.py --steps 4 --warmup 1 --batch-per-gpu 8 --config midi > $O/${R}_bench_midi_b8.json 2> $O/bench_midi.err
python bench.py --stream --steps 16 --warmup 4 > $O/${R}_bench_stream.json 2> $O/bench_stream.err
AFTER_GEMM_X6=1 python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/${R}_experiment_x6_bench_b8.json 2> $O/bench_x6.err
python scripts/time_codec.py --rounds 30 2>/dev/null | grep workload > $O/${R}_codec.jsonl
python scripts/time_encoders.py 2>/dev/null | grep workload >> $O/${R}_codec.jsonl
for b in 1 8; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st$b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --batch-per-gpu $b --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/st$b.log 2>&1)
  f=$(find $O/st$b -name "*kernel_stats.csv" | head -1)
  head -41 "$f" | cut -c1-260 > $O/${R}_bench_base_b${b}_kernel_stats.csv
  rm -rf $O/st$b
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python $GRAFT_REPO_ROOT/scripts/time_codec.py --rounds 3 --batches $b --only decode > $GRAFT_REPO_ROOT/$O/tr.log 2>&1)
  python scripts/trace_reduce.py $O/tr --end pqmf_inverse --rows > $O/${R}_decode_trace_b$b.jsonl
  rm -rf $O/tr
done
python scripts/pmc_run.py $O/${R}_pmc_codec_b1.json -- python scripts/time_codec.py --batches 1 --rounds 3 > $O/pmc_codec.log 2>&1
ls -la $O profiles | head -60
