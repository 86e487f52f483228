#!/bin/bash
# round 5 GPU passes: gpurun -- bash scripts/r5_pass.sh <pass>
set -u
pass=${1:?pass}
out=gpurun_out/r5_$pass
mkdir -p "$out"
export TMPDIR=/tmp
case "$pass" in
tests_a)  # everything that touches the changed code paths
    timeout 2400 python -m pytest tests/test_sample_clip_gpu.py tests/test_persist_protocol_gpu.py tests/test_sample_persist_gpu.py \
        tests/test_denoiser_gpu.py tests/test_baseline_size_gpu.py -x -q > "$out/tests.log" 2>&1
    tail -n 12 "$out/tests.log"
    ;;
tests_b)
    timeout 2400 python -m pytest tests/test_autoencoder_gpu.py tests/test_encoder_stream_gpu.py tests/test_streamer_gpu.py \
        tests/test_encoders_gpu.py tests/test_stream_persist_gpu.py tests/test_cabi_errors_gpu.py -x -q > "$out/tests.log" 2>&1
    tail -n 12 "$out/tests.log"
    ;;
tests)  # the whole -m gpu suite + smoke
    timeout 3400 python -m pytest tests -m gpu -x -q > "$out/tests.log" 2>&1
    tail -n 15 "$out/tests.log"
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
    tail -n 3 "$out/smoke.log"
    ;;
bench)
    timeout 900 python bench.py > "$out/b1.json" 2> "$out/b1.err"; tail -c 3000 "$out/b1.json"; tail -n 3 "$out/b1.err"
    timeout 900 python bench.py --batch-per-gpu 8 --steps 5 > "$out/b8.json" 2> "$out/b8.err"; tail -c 3000 "$out/b8.json"; tail -n 3 "$out/b8.err"
    ;;
*) echo "unknown pass"; exit 2;;
esac
