#!/bin/bash
# GPU passes of round 6 (gpurun -- bash scripts/r6_pass.sh <name>); output under gpurun_out/r6_<name>/
set -u
pass=${1:?pass}
O=gpurun_out/r6_$pass
mkdir -p $O
export TMPDIR=/tmp
case "$pass" in
first)  # the bench line with all BASELINE configs as legs, the dataset-embedding test, codec at 16 / 32 clips
    timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1.json 2> $O/bench_b1.err; echo "bench rc $?"
    python -c "
import json; d=json.load(open('$O/bench_b1.json')); print(d['ms_per_step'], d['value'], d['config']['sampler_path']); print(json.dumps(d.get('legs'), indent=1)[:3000])"
    timeout 900 python -m pytest tests/test_dataset_embed_gpu.py tests/test_denoiser_gpu.py tests/test_autoencoder_gpu.py -x -q 2>&1 | tail -5
    timeout 600 python scripts/time_codec.py --rounds 20 --batches 1,8,16,32 2>/dev/null | grep workload | tee $O/codec.jsonl
    ;;
gstag)  # experiment: row-tile groups of an XCD out of phase in the MLP-up phase -- do the output bursts shorten?
    for gs in -1 300 600 1000 1500; do
        echo "== AFTER_CLIP_GSTAG=$gs" >> $O/gstag.txt
        AFTER_CLIP_GSTAG=$gs timeout 300 python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids | grep "MLP-up\|L5 up\|L5 down\|step per" >> $O/gstag.txt
    done
    cat $O/gstag.txt
    ;;
h3)  # the two-piece fp16 form of the batch sampler's Linears: parity, same-box A/B against the three bf16 planes, per-phase trace
    timeout 1200 python -m pytest tests/test_sample_clip_gpu.py -x -q 2>&1 | tail -15
    for rep in 1 2; do
      for cfg in "base 8" "midi 8"; do set -- $cfg
        AFTER_CLIP_SPLIT=bf16 python scripts/time_sampler.py $1 $2 50 3 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/bf16 x 3 planes: /" | tee -a $O/ab_split.txt
        python scripts/time_sampler.py $1 $2 50 3 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/fp16 x 2 pieces: /" | tee -a $O/ab_split.txt
      done
    done
    python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids | tee $O/clip_step_trace.txt | tail -12
    ;;
h3acc)  # accuracy of the two-piece form vs fp64, the tests that run the batch sampler at BASELINE's sizes, the whole GPU suite
    timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -s -k "h3_two_piece" 2>&1 | grep "h3 \|passed\|failed\|Error" | tee $O/h3_accuracy.txt
    timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
    ;;
seg3)  # the two-piece fp16 form in the one-clip sampler: parity, same-box A/B (base and tiny), the bench line
    timeout 1200 python -m pytest tests/test_sample_persist_gpu.py tests/test_denoiser_gpu.py tests/test_persist_protocol_gpu.py -x -q 2>&1 | tail -6
    for rep in 1 2; do
      for cfg in base tiny; do
        AFTER_SEG_SPLIT=bf16 python scripts/time_sampler.py $cfg 1 50 7 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/bf16 x 3 planes: /" | tee -a $O/ab_split.txt
        python scripts/time_sampler.py $cfg 1 50 7 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/fp16 x 2 pieces: /" | tee -a $O/ab_split.txt
      done
    done
    python scripts/stream_step_trace.py --offline 2>/dev/null | grep -v amdgpu.ids | tee $O/offline_step_trace.txt | tail -8
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err; echo "bench rc $?"
    python -c "
import json; d=json.load(open('$O/bench_b1.json')); print(d['ms_per_step'], d['value'], d['config']['sampler_path'])
for k,v in d['legs'].items():
    if isinstance(v,dict): print(k, v.get('ms_per_step'), v.get('value'), v.get('sampler_path'), (v.get('roofline') or {}).get('frac'), v.get('error'))"
    ;;
tune1)  # small batches on the batch kernel vs launches; per-wave cycle counters of the fp16 K loops; MLP-down ring depth 4
    for b in 2 3 4; do
      AFTER_SAMPLE_CLIP=0 timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/launch /" | tee -a $O/clip_threshold.txt
      AFTER_SAMPLE_CLIP_MINB=2 timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/clip   /" | tee -a $O/clip_threshold.txt
    done
    cp after_amd/lib/libafter_hip.so $O/default.so
    cp scripts/variants/prof/libafter_hip.so after_amd/lib/libafter_hip.so
    python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids | tail -12 | tee $O/prof_qkv.txt
    AFTER_CLIP_GSTAG=-1 python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids | tail -12 | tee $O/prof_up.txt
    cp scripts/variants/dn4/libafter_hip.so after_amd/lib/libafter_hip.so
    python scripts/time_sampler.py base 8 50 3 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/dn4: /" | tee -a $O/variants.txt
    python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids | grep "L5\|step per" | tee -a $O/variants.txt
    cp $O/default.so after_amd/lib/libafter_hip.so; rm $O/default.so
    python scripts/time_sampler.py base 8 50 3 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/default: /" | tee -a $O/variants.txt
    ;;
suite)  # the whole GPU suite, the bench line with its legs, small-batch timings
    timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
    timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1.json 2> $O/bench_b1.err; echo "bench rc $?"; tail -3 $O/bench_b1.err
    python -c "
import json; d=json.load(open('$O/bench_b1.json')); print(d['ms_per_step'], d['value'], d['config']['sampler_path'], d['roofline']['frac'], d['roofline']['peak']); print(d['dtype'][:300])
for k,v in d['legs'].items():
    if isinstance(v,dict): print(k, v.get('ms_per_step'), v.get('value'), v.get('sampler_path'), (v.get('roofline') or {}).get('frac'), v.get('error'))"
    for b in 1 2 3 4 5 8; do
      timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | tee -a $O/clip_threshold.txt
    done
    ;;
wpre)  # the next GEMM phase's first weight slabs requested inside the barrier: parity, same-box A/B (AFTER_STEP_DBG bit 9 = off), trace; the rest of the suite
    timeout 1500 python -m pytest tests/test_sample_clip_gpu.py tests/test_sample_persist_gpu.py -x -q 2>&1 | tail -5
    for rep in 1 2; do
      AFTER_STEP_DBG=512 python scripts/time_sampler.py base 8 50 3 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/no prefetch: /" | tee -a $O/ab_wpre.txt
      python scripts/time_sampler.py base 8 50 3 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/prefetch:    /" | tee -a $O/ab_wpre.txt
    done
    python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids | tee $O/clip_step_trace.txt | tail -12
    timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_sample_clip_gpu.py --deselect tests/test_gemm_gpu.py --deselect tests/test_conv_tm_gpu.py 2>&1 | tail -6 | tee $O/gpu_suite.txt
    ;;
stream3)  # the streaming sampler's Linears on two-piece fp16 operands: parity, same-box A/B (AFTER_STREAM_SPLIT=fp32), per-phase trace
    timeout 1800 python -m pytest tests/test_stream_persist_gpu.py tests/test_streamer_gpu.py tests/test_persist_protocol_gpu.py tests/test_sample_clip_gpu.py tests/test_sample_persist_gpu.py -x -q 2>&1 | tail -5
    timeout 900 python -m pytest tests/test_baseline_size_gpu.py -x -q -k "streamer" 2>&1 | tail -3
    for rep in 1 2; do
      AFTER_STREAM_SPLIT=fp32 python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 MFMA chain:  ', d['ms_per_step'], 'ms per chunk', d['value'], 'xRT')" | tee -a $O/ab_stream_split.txt
      python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp16 x 2 pieces:  ', d['ms_per_step'], 'ms per chunk', d['value'], 'xRT')" | tee -a $O/ab_stream_split.txt
    done
    python scripts/stream_step_trace.py 2>/dev/null | grep -v amdgpu.ids | tee $O/stream_step_trace.txt | tail -14
    ;;
scales)  # the two-piece form's scales on weights far from the random-init scale; the fall-back to three bf16 planes
    timeout 900 python -m pytest tests/test_sample_persist_gpu.py -x -q -k "two_piece" 2>&1 | tail -15
    ;;
segslots)  # register-slot depth of the one-clip kernel's operand rings (k-blocks of a wave in flight), two-piece form
    cp after_amd/lib/libafter_hip.so $O/default.so
    for rep in 1 2; do
      for v in default s32 s23 s33 s44; do
        if [ $v = default ]; then cp $O/default.so after_amd/lib/libafter_hip.so; else cp scripts/variants/$v/libafter_hip.so after_amd/lib/libafter_hip.so; fi
        for cfg in base tiny; do
          python scripts/time_sampler.py $cfg 1 50 7 2>/dev/null | tail -1 | cut -c1-80 | sed "s/^/$v: /" | tee -a $O/seg_slots.txt
        done
      done
    done
    cp $O/default.so after_amd/lib/libafter_hip.so; rm $O/default.so
    ;;
final)  # the whole GPU suite on the final sources
    timeout 3400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
    ;;
grp)  # row-tile groups of the batch kernel free-running inside a layer (8-way barriers, per-layer hand-over rows, start delay): parity, A/B, trace
    cp after_amd/lib/libafter_hip.so $O/default.so
    cp scripts/variants/grp/libafter_hip.so after_amd/lib/libafter_hip.so
    timeout 1500 python -m pytest tests/test_sample_clip_gpu.py -x -q 2>&1 | tail -4
    timeout 900 python -m pytest tests/test_baseline_size_gpu.py -x -q -k "b8 or midi_b8" 2>&1 | tail -3
    for rep in 1 2; do
      cp $O/default.so after_amd/lib/libafter_hip.so
      python scripts/time_sampler.py base 8 50 3 2>/dev/null | tail -1 | cut -c1-80 | sed "s/^/HEAD (XCD-wide barriers, shared hand-over line):   /" | tee -a $O/ab_grouped.txt
      cp scripts/variants/grp/libafter_hip.so after_amd/lib/libafter_hip.so
      AFTER_CLIP_GROUPED=0 python scripts/time_sampler.py base 8 50 3 2>/dev/null | tail -1 | cut -c1-80 | sed "s/^/XCD-wide barriers, hand-over words on own lines:   /" | tee -a $O/ab_grouped.txt
      for d in 0 100 200 300 500 800; do
        AFTER_CLIP_GDELAY=$d python scripts/time_sampler.py base 8 50 3 2>/dev/null | tail -1 | cut -c1-80 | sed "s/^/group barriers, start delay $d x 10 ns per group:   /" | tee -a $O/ab_grouped.txt
      done
    done
    AFTER_CLIP_GDELAY=300 python scripts/time_sampler.py midi 8 50 3 2>/dev/null | tail -1 | cut -c1-80 | sed "s/^/midi, group barriers, delay 300: /" | tee -a $O/ab_grouped.txt
    AFTER_CLIP_GDELAY=300 python scripts/time_sampler.py base 4 50 3 2>/dev/null | tail -1 | cut -c1-80 | sed "s/^/4 clips, group barriers, delay 300: /" | tee -a $O/ab_grouped.txt
    python scripts/stream_step_trace.py --offline --clips 8 --xcd 3 2>/dev/null | grep -v amdgpu.ids | tee $O/clip_step_trace_grouped.txt | tail -22
    cp $O/default.so after_amd/lib/libafter_hip.so; rm $O/default.so
    ;;
m0)  # the LDS-DMA sites through __builtin_amdgcn_global_load_lds (compiler-managed M0) instead of inline asm: parity, same-box A/B
    timeout 2400 python -m pytest tests/test_gemm_gpu.py tests/test_conv_tm_gpu.py tests/test_sample_clip_gpu.py tests/test_sample_persist_gpu.py tests/test_autoencoder_gpu.py tests/test_stream_persist_gpu.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
    cp after_amd/lib/libafter_hip.so $O/builtin.so
    for rep in 1 2; do
      for v in builtin asm; do
        if [ $v = asm ]; then cp scripts/variants/m0asm/libafter_hip.so after_amd/lib/libafter_hip.so; else cp $O/builtin.so after_amd/lib/libafter_hip.so; fi
        for cfg in "base 1 7" "base 8 3" "tiny 1 7"; do set -- $cfg
          python scripts/time_sampler.py $1 $2 50 $3 2>/dev/null | tail -1 | cut -c1-100 | sed "s/^/$v $1 $2: /" | tee -a $O/ab.txt
        done
        timeout 600 python scripts/time_codec.py --rounds 20 --batches 1,8 2>/dev/null | grep workload | cut -c1-200 | sed "s/^/$v codec: /" | tee -a $O/ab.txt
      done
    done
    cp $O/builtin.so after_amd/lib/libafter_hip.so; rm -f $O/builtin.so
    timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/gpu_suite.txt
    ;;
pair)  # two clips per launch of the one-clip kernel (StepArgs::nclip, sample_seg_kernel<12, 512, 2>): parity, timings at 1 - 5 clips
    timeout 2400 python -m pytest tests/test_sample_persist_gpu.py -x -q 2>&1 | tail -8 | tee $O/tests.txt
    for b in 1 2 3 4 5; do
      AFTER_SEG_PAIR=0 timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/one per launch | batch kernel from 3:  /" | tee -a $O/pair.txt
      timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/pairs to 2:                            /" | tee -a $O/pair.txt
      AFTER_SAMPLE_SEG_PAIR_MAXB=5 timeout 300 python scripts/time_sampler.py base $b 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/pairs to 5:                            /" | tee -a $O/pair.txt
    done
    AFTER_T=128 timeout 300 python scripts/time_sampler.py base 2 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/T = 128, pair:  /" | tee -a $O/pair.txt
    AFTER_T=128 AFTER_SEG_PAIR=0 timeout 300 python scripts/time_sampler.py base 2 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/T = 128, one per launch:  /" | tee -a $O/pair.txt
    timeout 300 python scripts/time_sampler.py midi 2 50 3 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/midi pair:  /" | tee -a $O/pair.txt
    ;;
pairtrace)  # per-phase stamps of the one-clip kernel with one and with two clips per launch
    python scripts/stream_step_trace.py --offline --clips 1 2>/dev/null | grep -v amdgpu.ids | tee $O/one.txt | tail -14
    python scripts/stream_step_trace.py --offline --clips 2 2>/dev/null | grep -v amdgpu.ids | tee $O/two.txt | tail -50
    ;;
stagger)  # experiment (reverted): one-clip kernel, halo chunks last in item order, the neighbour's sequence word fetched under the qkv phase, XCD start stagger
    timeout 2400 python -m pytest tests/test_sample_persist_gpu.py tests/test_persist_protocol_gpu.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
    for rep in 1 2; do
      for st in 0 200 400 800 1200; do
        for cfg in "base 1" "base 2" "tiny 1"; do set -- $cfg
          AFTER_SEG_STAGGER=$st timeout 300 python scripts/time_sampler.py $1 $2 50 5 2>&1 | grep "sample " | cut -c1-70 | sed "s/^/stagger $st: /" | tee -a $O/stagger.txt
        done
      done
    done
    AFTER_SEG_STAGGER=800 python scripts/stream_step_trace.py --offline --clips 1 2>/dev/null | grep -v amdgpu.ids | tee $O/one_800.txt | tail -6
    AFTER_SEG_STAGGER=800 python scripts/stream_step_trace.py --offline --clips 2 2>/dev/null | grep -v amdgpu.ids | tee $O/two_800.txt | tail -6
    ;;
ubench)  # the stand-alone micro-benchmarks behind the design's price list (XCD barriers, halo hand-over, L2 round trips, DMA issue rate)
    mkdir -p gpurun_out/final_r6
    for u in xcd_barrier xcd_local xcd_halo xcd_barrier2 l2_rtt dma_issue; do
      [ -x scripts/ubench/$u.bin ] || hipcc --offload-arch=gfx950 -O3 scripts/ubench/$u.hip -o scripts/ubench/$u.bin
      timeout 300 ./scripts/ubench/$u.bin > gpurun_out/final_r6/r6_$u.jsonl 2> $O/$u.err; echo "$u rc $? $(wc -c < gpurun_out/final_r6/r6_$u.jsonl) bytes"
    done
    ;;
b2prof)  # kernel stats and matrix-pipe busy fraction of the pair launch
    mkdir -p gpurun_out/final_r6; F=gpurun_out/final_r6
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/ss2 -- python $GRAFT_REPO_ROOT/scripts/time_sampler.py base 2 50 5 > $GRAFT_REPO_ROOT/$O/ss2.log 2>&1)
    f=$(find $O/ss2 -name "*kernel_stats.csv" | head -1)
    head -21 "$f" | cut -c1-260 | tee $F/r6_sampler_b2_kernel_stats.csv | head -6
    rm -rf $O/ss2
    python bench.py --pmc-mfma --batch-per-gpu 2 > $O/mfma_b2.log 2>&1; tail -3 $O/mfma_b2.log | cut -c1-300
    cp profiles/r6_pmc_mfma_base_b2.json $F/ 2>/dev/null
    ;;
final2)  # final sources: the whole GPU suite, then every artefact of the round on the same lease
    timeout 3400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/gpu_suite.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
    bash scripts/round_profiles.sh r6 > $O/round_profiles.log 2>&1
    ;;
kvwarm)  # streaming sampler: idle workgroups of the LayerNorm phase warm the layer's K / V ring rows (and n sixteenths of the qkv weights)
    for rep in 1 2; do
      for wm in "0,16,4" "1,16,4" "2,16,4" "4,16,4"; do
        AFTER_STEP_WARM=$wm python bench.py --stream --steps 24 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('AFTER_STEP_WARM=$wm:', d['ms_per_step'], 'ms per chunk')" | tee -a $O/kvwarm.txt
      done
    done
    ;;
codech3)  # the codec's GroupNorm-bounded bf16-pipe convs on two fp16 pieces: parity (every codec test), same-box A/B, per-launch traces
    timeout 2400 python -m pytest tests/test_autoencoder_gpu.py tests/test_conv_tm_gpu.py tests/test_baseline_size_gpu.py tests/test_dataset_embed_gpu.py tests/test_checkpoint.py tests/test_large_sizes_gpu.py -x -q 2>&1 | tail -6
    for rep in 1 2; do
      AFTER_CONV_H3=0 python scripts/time_codec.py --rounds 20 --batches 1,8,32 2>/dev/null | grep workload | sed "s/^/three bf16 planes: /" | tee -a $O/ab_conv_h3.txt
      python scripts/time_codec.py --rounds 20 --batches 1,8,32 2>/dev/null | grep workload | sed "s/^/two fp16 pieces:   /" | tee -a $O/ab_conv_h3.txt
    done
    ;;
codech3b)  # the rest of the codec tests; does the two-piece form pay on MORE layers (AFTER_CONV_X6=2: wherever eligible)?
    timeout 2400 python -m pytest tests/test_baseline_size_gpu.py tests/test_dataset_embed_gpu.py tests/test_checkpoint.py tests/test_large_sizes_gpu.py tests/test_encoder_stream_gpu.py tests/test_streamer_gpu.py -x -q 2>&1 | tail -4
    for rep in 1 2; do
      python scripts/time_codec.py --rounds 20 --batches 1,2,8,32 2>/dev/null | grep workload | sed "s/^/by size (default):   /" | tee -a $O/ab_conv_h3_wide.txt
      AFTER_CONV_X6=2 python scripts/time_codec.py --rounds 20 --batches 1,2,8,32 2>/dev/null | grep workload | sed "s/^/wherever eligible:   /" | tee -a $O/ab_conv_h3_wide.txt
    done
    AFTER_AE_TRACE=1 AFTER_CONV_X6=2 python scripts/time_codec.py --rounds 1 --batches 8 --only encode 2>&1 | grep "ae conv" | sort | uniq -c | sort -rn | head -40 | tee $O/encode_layers_x6_2.txt
    ;;
codech3c)  # the widened rule (filled launches of any width on two pieces): codec tests, codec timings, the bench line
    timeout 2400 python -m pytest tests/test_autoencoder_gpu.py tests/test_baseline_size_gpu.py tests/test_dataset_embed_gpu.py tests/test_checkpoint.py tests/test_large_sizes_gpu.py -x -q 2>&1 | tail -4
    for rep in 1 2; do
      AFTER_CONV_H3=0 python scripts/time_codec.py --rounds 20 --batches 1,2,8,16,32 2>/dev/null | grep workload | sed "s/^/three bf16 planes: /" | tee -a $O/ab_conv_h3_final.txt
      python scripts/time_codec.py --rounds 20 --batches 1,2,8,16,32 2>/dev/null | grep workload | sed "s/^/two fp16 pieces:   /" | tee -a $O/ab_conv_h3_final.txt
    done
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err; echo "bench rc $?"
    python -c "
import json; d=json.load(open('$O/bench_b1.json')); print(d['ms_per_step'], d['value'], d['config']['sampler_path'])
for k,v in d['legs'].items():
    if isinstance(v,dict): print(k, v.get('ms_per_step'), v.get('value'), v.get('sampler_path'), (v.get('roofline') or {}).get('frac'), v.get('error'))"
    ;;
esac
