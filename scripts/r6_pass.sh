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
esac
