#!/bin/bash
# Codec check after a conv_tm / act_pad change: parity tests of everything that runs on conv_tm, then the codec timings and
# per-launch traces (same commands as round_profiles.sh) into gpurun_out/codec_pass/.
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=gpurun_out/codec_pass; mkdir -p $O
python -m pytest tests/test_autoencoder_gpu.py tests/test_encoders_gpu.py tests/test_unet1d_gpu.py tests/test_encoder_stream_gpu.py \
  tests/test_streamer_gpu.py tests/test_conv_tm_gpu.py tests/test_properties_gpu.py -x -q 2>&1 | tail -5
python scripts/time_codec.py --rounds 30 2>/dev/null | grep workload > $O/codec.jsonl; cat $O/codec.jsonl
for b in 1 8; do
  for w in decode encode; do
    e=pqmf_inverse; [ $w = encode ] && e=pqmf_forward
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python $GRAFT_REPO_ROOT/scripts/time_codec.py --rounds 3 --batches $b --only $w > $GRAFT_REPO_ROOT/$O/tr.log 2>&1)
    python scripts/trace_reduce.py $O/tr --end $e --rows > $O/${w}_trace_b$b.jsonl
    rm -rf $O/tr
    head -c 400 $O/${w}_trace_b$b.jsonl; echo
  done
done
