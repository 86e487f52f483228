#!/bin/bash
# Regenerates the per-round measurement artefacts on the GPU box (run through gpurun from the repo root):
#   bash scripts/round_profiles.sh r3
# Writes into gpurun_out/final_<round>/; the summaries are then copied into profiles/ by hand.
R=${1:-r3}
O=gpurun_out/final_$R
mkdir -p $O
export TMPDIR=/tmp
python bench.py --pmc > $O/pmc_b1.log 2>&1   # first: the bench lines below quote their traffic figures
python bench.py --pmc --batch-per-gpu 8 > $O/pmc_b8.log 2>&1
python bench.py --pmc-mfma > $O/mfma_b1.log 2>&1   # matrix-pipe busy fraction per kernel (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE)
python bench.py --pmc-mfma --batch-per-gpu 8 > $O/mfma_b8.log 2>&1
cp profiles/${R}_pmc_hbm_base_b1.json profiles/${R}_pmc_hbm_base_b8.json profiles/${R}_pmc_mfma_base_b1.json profiles/${R}_pmc_mfma_base_b8.json $O/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_b1.json 2> $O/bench_b1.err   # (the driver's line: the headline + `legs` = the other BASELINE configs)
python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/${R}_bench_b8.json 2> $O/bench_b8.err
python bench.py --steps 4 --warmup 1 --batch-per-gpu 8 --config midi > $O/${R}_bench_midi_b8.json 2> $O/bench_midi.err
python bench.py --stream --steps 16 --warmup 4 > $O/${R}_bench_stream.json 2> $O/bench_stream.err
python bench.py --steps 5 --warmup 2 --from-audio --config tiny > $O/${R}_bench_tiny_from_audio.json 2> $O/bench_fa.err
python bench.py --steps 5 --warmup 2 --from-audio --no-cpu-baseline > $O/${R}_bench_base_from_audio.json 2>> $O/bench_fa.err
# same-box A/B of the GEMM paths and of the persistent many-row tile
for c in 0 1; do
  AFTER_GEMM_X6=$c python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'b1', 'AFTER_GEMM_X6': $c, 'ms_per_step': d['ms_per_step'], 'xrt': d['value']}))" >> $O/${R}_ab_gemm_path.jsonl
  AFTER_GEMM_X6=$c python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'b8', 'AFTER_GEMM_X6': $c, 'ms_per_step': d['ms_per_step'], 'xrt': d['value']}))" >> $O/${R}_ab_gemm_path.jsonl
done
for p in 1 0; do
  AFTER_GEMM_X6_PERSIST=$p python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'b8', 'AFTER_GEMM_X6_PERSIST': $p, 'ms_per_step': d['ms_per_step'], 'xrt': d['value']}))" >> $O/${R}_ab_gemm_path.jsonl
done
python scripts/bench_gemm_x6.py $O/${R}_gemm_x6_sweep.jsonl > $O/x6_sweep.log 2>&1
python scripts/time_codec.py --rounds 30 --batches 1,8,16,32 2>/dev/null | grep workload > $O/${R}_codec.jsonl   # (16 / 32: the dataset scripts' batches)
python scripts/time_encoders.py 2>/dev/null | grep workload >> $O/${R}_codec.jsonl
# round 4: the bf16-pipe convs of the decoder (conv_x6.hip): same-box A/B of the codec, per-layer sweep against the fp32 kernel
for m in 0 1 0 1; do
  AFTER_CONV_X6=$m python scripts/time_codec.py --rounds 30 2>/dev/null | grep workload | python -c "import json,sys
for l in sys.stdin:
    d=json.loads(l); d['AFTER_CONV_X6']=$m; print(json.dumps(d))" >> $O/${R}_ab_conv_x6.jsonl
done
X6L="dec0 k3 ,dec1 k3,dec1 k1,dec2 k3 ,dec2 k1,dec3 k3,dec3 k1,up2"
python scripts/bench_conv.py --x6 --tiles 0 --layers "$X6L" 2>/dev/null | grep layer > $O/${R}_bench_conv_x6_b1.jsonl
python scripts/bench_conv.py --batch 8 --x6 --tiles 0 --layers "$X6L" 2>/dev/null | grep layer > $O/${R}_bench_conv_x6_b8.jsonl
for u in xcd_barrier xcd_local xcd_halo xcd_barrier2 l2_rtt dma_issue; do  # (the stand-alone micro-benchmarks: built where they run)
  [ -x scripts/ubench/$u.bin ] || hipcc --offload-arch=gfx950 -O3 scripts/ubench/$u.hip -o scripts/ubench/$u.bin 2>> $O/ubench_build.err
done
./scripts/ubench/xcd_barrier.bin > $O/${R}_xcd_barrier.jsonl 2>/dev/null
./scripts/ubench/xcd_local.bin > $O/${R}_xcd_local.jsonl 2>/dev/null
./scripts/ubench/xcd_halo.bin > $O/${R}_xcd_halo.jsonl 2>/dev/null
./scripts/ubench/xcd_barrier2.bin > $O/${R}_xcd_barrier2.jsonl 2>/dev/null   # round 4: the barrier under spread arrivals, the L2 round trips, the DMA issue rate
./scripts/ubench/l2_rtt.bin > $O/${R}_l2_rtt.jsonl 2>/dev/null
./scripts/ubench/dma_issue.bin > $O/${R}_dma_issue.jsonl 2>/dev/null
AFTER_SAMPLE_PERSIST=0 python scripts/time_graph.py > $O/${R}_graph_vs_eager.jsonl 2>/dev/null   # (launch path: eager launches vs hipGraph replay)
AFTER_SAMPLE_PERSIST=0 python scripts/time_sampler.py base 1 50 7 2>/dev/null | tail -1 > $O/${R}_ab_sample_persist.txt
python scripts/time_sampler.py base 1 50 7 2>/dev/null | tail -1 >> $O/${R}_ab_sample_persist.txt
AFTER_SAMPLE_PERSIST=0 python scripts/time_sampler.py base 1 50 7 2>/dev/null | tail -1 >> $O/${R}_ab_sample_persist.txt
python scripts/time_sampler.py base 1 50 7 2>/dev/null | tail -1 >> $O/${R}_ab_sample_persist.txt
# round 5: the clip-per-XCD persistent sampler at 8 clips -- same-box A/B against the launch path, per-phase timeline; cost of the
# default checked mode of the offline persistent samplers (one event wait per call)
for rep in 1 2; do
  AFTER_SAMPLE_CLIP=0 python scripts/time_sampler.py base 8 50 5 2>/dev/null | tail -1 | sed 's/^/launch path:      /' >> $O/${R}_ab_sample_clip.txt
  python scripts/time_sampler.py base 8 50 5 2>/dev/null | tail -1 | sed 's/^/clip-per-XCD:     /' >> $O/${R}_ab_sample_clip.txt
done
for cfgb in "midi 8" "base 6" "base 16"; do set -- $cfgb
  AFTER_SAMPLE_CLIP=0 python scripts/time_sampler.py $1 $2 50 3 2>/dev/null | tail -1 | sed 's/^/launch path:      /' >> $O/${R}_ab_sample_clip.txt
  python scripts/time_sampler.py $1 $2 50 3 2>/dev/null | tail -1 | sed 's/^/clip-per-XCD:     /' >> $O/${R}_ab_sample_clip.txt
done
# round 5b: the attention inside the qkv tiles against the item form (AFTER_CLIP_FUSE), same box, interleaved; the opt-in bf16
# tolerance tier against the default arithmetic (sampler alone + the separate bench legs); tiny on the persistent one-clip sampler
for rep in 1 2; do
  for cfg in base midi; do
    AFTER_CLIP_FUSE=0 python scripts/time_sampler.py $cfg 8 50 3 2>/dev/null | tail -1 | cut -c1-90 | sed 's/^/qkv rows + items:      /' >> $O/${R}_ab_clip_fuse.txt
    python scripts/time_sampler.py $cfg 8 50 3 2>/dev/null | tail -1 | cut -c1-90 | sed 's/^/tiles attend in place: /' >> $O/${R}_ab_clip_fuse.txt
  done
done
for b in 1 8; do
  for m in 1 3 1 3; do
    AFTER_TIME_GEMM_PATH=$m python scripts/time_sampler.py base $b 50 5 2>/dev/null | tail -1 | cut -c1-90 | sed "s/^/gemm path $m: /" >> $O/${R}_ab_bf16_tier.txt
  done
done
# round 6: the Linears' arithmetic in the three persistent samplers -- two fp16 pieces (default) against three bf16 planes / the fp32 chain
for rep in 1 2; do
  for cfgb in "base 1" "tiny 1" "base 8" "midi 8"; do set -- $cfgb
    AFTER_SEG_SPLIT=bf16 AFTER_CLIP_SPLIT=bf16 python scripts/time_sampler.py $1 $2 50 5 2>/dev/null | tail -1 | cut -c1-90 | sed "s/^/bf16 x 3 planes: /" >> $O/${R}_ab_split.txt
    python scripts/time_sampler.py $1 $2 50 5 2>/dev/null | tail -1 | cut -c1-90 | sed "s/^/fp16 x 2 pieces: /" >> $O/${R}_ab_split.txt
  done
  AFTER_STREAM_SPLIT=fp32 python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 MFMA chain:  ', d['ms_per_step'], 'ms per chunk', d['value'], 'xRT')" >> $O/${R}_ab_stream_split.txt
  python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp16 x 2 pieces:  ', d['ms_per_step'], 'ms per chunk', d['value'], 'xRT')" >> $O/${R}_ab_stream_split.txt
done
for b in 1 2 3 4 5 8; do timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 >> $O/${R}_clip_threshold_shipped.txt; done
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -s -k "h3_two_piece" 2>&1 | grep "h3 \|passed\|failed" > $O/${R}_h3_accuracy.txt
python bench.py --bf16-tier --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_bf16_tier_b1.json 2> $O/bench_tier.err
python bench.py --bf16-tier --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/${R}_bench_bf16_tier_b8.json 2>> $O/bench_tier.err
for p in 0 1 0 1; do
  AFTER_SAMPLE_PERSIST=$p python scripts/time_sampler.py tiny 1 50 7 2>/dev/null | tail -1 | cut -c1-90 | sed "s/^/AFTER_SAMPLE_PERSIST=$p: /" >> $O/${R}_ab_tiny_persist.txt
done
AFTER_T=192 python scripts/time_sampler.py base 1 50 5 2>/dev/null | tail -1 | cut -c1-90 | sed 's/^/T=192 persistent (6 segments): /' >> $O/${R}_ab_tiny_persist.txt
AFTER_T=192 AFTER_SAMPLE_PERSIST=0 python scripts/time_sampler.py base 1 50 5 2>/dev/null | tail -1 | cut -c1-90 | sed 's/^/T=192 launches:                /' >> $O/${R}_ab_tiny_persist.txt
AFTER_T=64 python scripts/time_sampler.py base 1 50 5 2>/dev/null | tail -1 | cut -c1-90 | sed 's/^/T=64 persistent (4 segments):  /' >> $O/${R}_ab_tiny_persist.txt
AFTER_T=64 AFTER_SAMPLE_PERSIST=0 python scripts/time_sampler.py base 1 50 5 2>/dev/null | tail -1 | cut -c1-90 | sed 's/^/T=64 launches:                 /' >> $O/${R}_ab_tiny_persist.txt
python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids > $O/${R}_clip_step_trace.txt
python scripts/persist_check_cost.py > $O/${R}_persist_check_cost.jsonl 2>/dev/null
python scripts/seg_context.py 2>/dev/null | tail -9 > $O/${R}_clip_breakdown_b1.txt
python scripts/stream_step_trace.py --offline > $O/${R}_offline_step_trace.txt 2>/dev/null
# two clips per launch of the one-clip kernel: per-phase stamps, same-box A/B against one clip per launch (T = 256 and 128, midi)
python scripts/stream_step_trace.py --offline --clips 2 2>/dev/null | grep -v amdgpu.ids > $O/${R}_pair_step_trace.txt
for rep in 1 2; do
  for cfgt in "base 256" "base 128" "midi 256"; do set -- $cfgt
    AFTER_T=$2 AFTER_SEG_PAIR=0 python scripts/time_sampler.py $1 2 50 5 2>/dev/null | tail -1 | cut -c1-75 | sed "s/^/T = $2, one clip per launch: /" >> $O/${R}_ab_pair.txt
    AFTER_T=$2 python scripts/time_sampler.py $1 2 50 5 2>/dev/null | tail -1 | cut -c1-75 | sed "s/^/T = $2, the pair in one:      /" >> $O/${R}_ab_pair.txt
  done
done
# the persistent streaming step: same-box A/B against the launch path, per-phase timeline, kernel stats of a chunk
for p in 1 0 1 0; do
  AFTER_STREAM_PERSIST=$p python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'stream 8x100 steps', 'AFTER_STREAM_PERSIST': $p, 'ms_per_chunk': d['ms_per_step'], 'xrt': d['value']}))" >> $O/${R}_ab_stream_persist.jsonl
done
python scripts/stream_step_trace.py > $O/${R}_stream_step_trace.txt 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/sp -- python $GRAFT_REPO_ROOT/bench.py --stream --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/sp.log 2>&1)
f=$(find $O/sp -name "*kernel_stats.csv" | head -1)
head -31 "$f" | cut -c1-260 > $O/${R}_bench_stream_kernel_stats.csv
rm -rf $O/sp
for b in 1 8; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st$b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --batch-per-gpu $b --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/st$b.log 2>&1)
  f=$(find $O/st$b -name "*kernel_stats.csv" | head -1)
  head -41 "$f" | cut -c1-260 > $O/${R}_bench_base_b${b}_kernel_stats.csv
  rm -rf $O/st$b
  # the sampler alone (default GEMM path only: bench.py's roofline legs also run the fp32 MFMA path once)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/ss$b -- python $GRAFT_REPO_ROOT/scripts/time_sampler.py base $b 50 5 > $GRAFT_REPO_ROOT/$O/ss$b.log 2>&1)
  f=$(find $O/ss$b -name "*kernel_stats.csv" | head -1)
  head -21 "$f" | cut -c1-260 > $O/${R}_sampler_b${b}_kernel_stats.csv
  rm -rf $O/ss$b
  if [ $b = 1 ]; then  # ... and two clips in one launch of the one-clip kernel
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/ss2 -- python $GRAFT_REPO_ROOT/scripts/time_sampler.py base 2 50 5 > $GRAFT_REPO_ROOT/$O/ss2.log 2>&1)
    f=$(find $O/ss2 -name "*kernel_stats.csv" | head -1)
    head -21 "$f" | cut -c1-260 > $O/${R}_sampler_b2_kernel_stats.csv
    rm -rf $O/ss2
  fi
  for w in decode encode; do
    e=pqmf_inverse; [ $w = encode ] && e=pqmf_forward
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python $GRAFT_REPO_ROOT/scripts/time_codec.py --rounds 3 --batches $b --only $w > $GRAFT_REPO_ROOT/$O/tr.log 2>&1)
    python scripts/trace_reduce.py $O/tr --end $e --rows > $O/${R}_${w}_trace_b$b.jsonl
    rm -rf $O/tr
  done
done
python scripts/pmc_run.py $O/${R}_pmc_codec_b1.json -- python scripts/time_codec.py --batches 1 --rounds 3 > $O/pmc_codec.log 2>&1
for b in 1 8; do python scripts/pmc_run.py --mfma $O/${R}_pmc_mfma_codec_b$b.json -- python scripts/time_codec.py --batches $b --rounds 3 >> $O/pmc_codec.log 2>&1; done   # matrix-pipe busy fraction of the conv kernels
# one whole chunk of the streaming path, launch by launch (everything between two launches of the persistent sampler)
bash scripts/stream_chunk_trace.sh > $O/stream_chunk.log 2>&1; cp gpurun_out/stream_chunk/chunk_trace.jsonl $O/${R}_stream_chunk_trace.jsonl; cp gpurun_out/stream_chunk/time_stream.json $O/${R}_time_stream.json
ls -la $O | head -60
