#!/bin/bash
# The GPU passes of a round as they were run through `gpurun -- bash scripts/gpu_pass.sh <pass>` (one lease each).
# Output goes to gpurun_out/<pass>/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
set -u
pass=${1:?pass name}
out=gpurun_out/$pass
mkdir -p "$out"
export TMPDIR=/tmp
ts() { # ts <label> <env...> : 50-step base B=1 sampler time under an environment
    local label=$1; shift
    env "$@" timeout 300 python scripts/time_sampler.py base 1 50 7 2>&1 | grep "sample " | sed "s/^/$label: /" >> "$out/times.log"
}
case "$pass" in
seg_a)  # round 4: persistent offline sampler after the de-spill pass -- K split variants x L2 warming variants
    timeout 900 python -m pytest tests/test_sample_persist_gpu.py -x -q > "$out/test_k0.log" 2>&1
    AFTER_SEG_K8=2 AFTER_SEG_W=15 timeout 900 python -m pytest tests/test_sample_persist_gpu.py -x -q > "$out/test_k2w15.log" 2>&1
    AFTER_SEG_K8=3 AFTER_SEG_W=10 timeout 900 python -m pytest tests/test_sample_persist_gpu.py -x -q > "$out/test_k3w10.log" 2>&1
    tail -3 "$out"/test_*.log
    ts launch AFTER_SAMPLE_PERSIST=0
    for k in 0 2 3; do
        for w in 0 1 2 4 8 10 15; do
            ts "persist k8=$k w=$w" AFTER_SAMPLE_PERSIST=1 AFTER_SEG_K8=$k AFTER_SEG_W=$w
        done
    done
    ts "persist k8=0 w=0 warm0,16,4" AFTER_SAMPLE_PERSIST=1 AFTER_SEG_WARM=0,16,4
    ts "persist k8=0 w=10 warm0,16,0" AFTER_SAMPLE_PERSIST=1 AFTER_SEG_W=10 AFTER_SEG_WARM=0,16,0
    ts launch AFTER_SAMPLE_PERSIST=0
    cat "$out/times.log"
    for k in 0 2; do
        for w in 0 15; do
            AFTER_SEG_K8=$k AFTER_SEG_W=$w timeout 300 python scripts/stream_step_trace.py --offline > "$out/trace_k${k}_w${w}.txt" 2>&1
        done
    done
    tail -12 "$out/trace_k0_w0.txt"
    ;;
vtrace)  # per-phase traces of the kernel variants
    cp after_amd/lib/libafter_hip.so "$out/default.so"
    for v in default $(ls scripts/variants); do
        if [ "$v" = default ]; then cp "$out/default.so" after_amd/lib/libafter_hip.so; else cp "scripts/variants/$v/libafter_hip.so" after_amd/lib/libafter_hip.so; fi
        timeout 300 python scripts/stream_step_trace.py --offline --xcd 3 > "$out/trace_$v.txt" 2>&1
        echo "== $v"; sed -n 4,9p "$out/trace_$v.txt"; tail -4 "$out/trace_$v.txt"
    done
    cp "$out/default.so" after_amd/lib/libafter_hip.so; rm "$out/default.so"
    ;;
variants)  # A/B of the kernel variants built by scripts/build_variant.sh (each a complete libafter_hip.so), same box, interleaved
    cp after_amd/lib/libafter_hip.so "$out/default.so"
    for rep in 1 2; do
        for v in default $(ls scripts/variants); do
            if [ "$v" = default ]; then cp "$out/default.so" after_amd/lib/libafter_hip.so; else cp "scripts/variants/$v/libafter_hip.so" after_amd/lib/libafter_hip.so; fi
            ts "$v" AFTER_X=1
            if [ $rep = 1 ] && [ "${VARIANT_TESTS:-1}" = 1 ]; then timeout 600 python -m pytest tests/test_sample_persist_gpu.py -x -q 2>&1 | tail -1 | sed "s/^/$v tests: /" >> "$out/times.log"; fi
        done
    done
    cp "$out/default.so" after_amd/lib/libafter_hip.so; rm "$out/default.so"
    sort "$out/times.log"
    ;;
seg_c)  # round 4: halo keys last in the offline persistent sampler's attention
    timeout 900 python -m pytest tests/test_sample_persist_gpu.py tests/test_persist_protocol_gpu.py -x -q > "$out/test.log" 2>&1
    tail -n 3 "$out"/test.log
    ts launch AFTER_SAMPLE_PERSIST=0
    ts persist; ts persist
    cat "$out/times.log"
    timeout 300 python scripts/stream_step_trace.py --offline > "$out/trace.txt" 2>&1
    timeout 300 python scripts/stream_step_trace.py --offline --xcd 3 > "$out/trace_xcd3.txt" 2>&1
    head -12 "$out/trace_xcd3.txt"; tail -5 "$out/trace.txt"
    ;;
seg_b)  # round 4: attention of the offline persistent sampler with one round trip per item; weight prefetch across the barriers
    timeout 900 python -m pytest tests/test_sample_persist_gpu.py -x -q > "$out/test_pre0.log" 2>&1
    AFTER_SEG_PRE=7 timeout 900 python -m pytest tests/test_sample_persist_gpu.py -x -q > "$out/test_pre7.log" 2>&1
    tail -n 3 "$out"/test_*.log
    ts launch AFTER_SAMPLE_PERSIST=0
    for p in 0 1 6 7 14 15 0; do ts "persist pre=$p" AFTER_SEG_PRE=$p; done
    cat "$out/times.log"
    for p in 0 7; do AFTER_SEG_PRE=$p timeout 300 python scripts/stream_step_trace.py --offline > "$out/trace_pre$p.txt" 2>&1; done
    head -12 "$out/trace_pre0.txt"; tail -5 "$out/trace_pre0.txt"
    ;;
persist_tests)  # round 4: the persistent samplers' host protocol + every test that touches them
    timeout 1700 python -m pytest tests/test_persist_protocol_gpu.py tests/test_stream_persist_gpu.py tests/test_sample_persist_gpu.py \
        tests/test_baseline_size_gpu.py tests/test_denoiser_gpu.py tests/test_streamer_gpu.py -x -q > "$out/tests.log" 2>&1
    tail -30 "$out/tests.log"
    ts launch AFTER_SAMPLE_PERSIST=0
    ts persist_default
    cat "$out/times.log"
    ;;
counters)  # which MFMA / busy counters this rocprofv3 knows on gfx950
    cd /tmp; rocprofv3 --list-avail 2>&1 | grep -i -B1 -A3 "mfma\|SQ_BUSY_CY\|SQ_WAVES \|GRBM_GUI_ACTIVE\|SQ_WAVE_CYCLES\|VALUBusy\|MfmaUtil" | head -150 > "$GRAFT_REPO_ROOT/$out/avail.txt" 2>&1
    cd "$GRAFT_REPO_ROOT"; head -150 "$out/avail.txt"
    ;;
pmc)  # HBM traffic + MFMA-busy counter passes of the bench command at B = 1 and B = 8 (written into profiles/ -> copied to $out)
    for b in 1 8; do
        timeout 900 python bench.py --pmc --batch-per-gpu $b > "$out/pmc_b$b.log" 2>&1; tail -2 "$out/pmc_b$b.log"
        timeout 900 python bench.py --pmc-mfma --batch-per-gpu $b > "$out/mfma_b$b.log" 2>&1; tail -2 "$out/mfma_b$b.log"
    done
    cp profiles/r4_pmc_* "$out/" 2>/dev/null
    ;;
bench)  # the default bench line (B = 1) + B = 8
    timeout 900 python bench.py > "$out/b1.json" 2> "$out/b1.err"; tail -c 6000 "$out/b1.json"
    timeout 900 python bench.py --batch-per-gpu 8 --steps 5 > "$out/b8.json" 2> "$out/b8.err"; tail -c 1500 "$out/b8.json"
    ;;
tests)  # the whole -m gpu suite + smoke
    timeout 3000 python -m pytest tests -m gpu -x -q > "$out/tests.log" 2>&1
    tail -15 "$out/tests.log"
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
    tail -3 "$out/smoke.log"
    ;;
vq)  # variants: sampler time only, interleaved (no tests) + traces
    cp after_amd/lib/libafter_hip.so "$out/default.so"
    for rep in 1 2; do
        for v in default $(ls scripts/variants); do
            if [ "$v" = default ]; then cp "$out/default.so" after_amd/lib/libafter_hip.so; else cp "scripts/variants/$v/libafter_hip.so" after_amd/lib/libafter_hip.so; fi
            ts "$v" AFTER_X=1
        done
    done
    for v in default $(ls scripts/variants); do
        if [ "$v" = default ]; then cp "$out/default.so" after_amd/lib/libafter_hip.so; else cp "scripts/variants/$v/libafter_hip.so" after_amd/lib/libafter_hip.so; fi
        timeout 300 python scripts/stream_step_trace.py --offline --xcd 3 > "$out/trace_$v.txt" 2>&1
        echo "== $v"; sed -n 4,9p "$out/trace_$v.txt" | cut -c1-48; tail -5 "$out/trace_$v.txt" | head -3
    done
    cp "$out/default.so" after_amd/lib/libafter_hip.so; rm "$out/default.so"
    sort "$out/times.log" | cut -c1-80
    ;;
quick)  # the persistent samplers' own tests + sampler time + per-phase trace (the inner loop of a kernel change)
    timeout 900 python -m pytest tests/test_sample_persist_gpu.py tests/test_stream_persist_gpu.py tests/test_persist_protocol_gpu.py -x -q > "$out/test.log" 2>&1
    tail -n 3 "$out"/test.log
    ts persist; ts persist
    cat "$out/times.log"
    timeout 300 python scripts/stream_step_trace.py --offline --xcd 3 --detail > "$out/trace_xcd3.txt" 2>&1
    grep -v amdgpu.ids "$out/trace_xcd3.txt" | cut -c1-48 | sed -n 2,14p; tail -6 "$out/trace_xcd3.txt"
    timeout 300 python bench.py --stream --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stream', d['value'], d['ms_per_step'], d['config'].get('sampler_path'))"
    ;;
*)
    echo "unknown pass $pass"; exit 2;;
esac
