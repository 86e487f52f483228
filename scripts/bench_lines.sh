#!/bin/bash
# The bench lines of a round on one lease (after a change to bench.py or the codec: the sampler's counter passes stay valid while the
# hashed kernel sources do): bash scripts/bench_lines.sh r5 -> gpurun_out/lines_r5/
R=${1:-r5}; O=gpurun_out/lines_$R; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_b1.json 2> $O/b1.err
python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/${R}_bench_b8.json 2> $O/b8.err
python bench.py --steps 4 --warmup 1 --batch-per-gpu 8 --config midi > $O/${R}_bench_midi_b8.json 2> $O/midi.err
python bench.py --steps 5 --warmup 2 --from-audio --config tiny > $O/${R}_bench_tiny_from_audio.json 2> $O/fa.err
python bench.py --steps 5 --warmup 2 --from-audio --no-cpu-baseline > $O/${R}_bench_base_from_audio.json 2>> $O/fa.err
python bench.py --bf16-tier --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_bench_bf16_tier_b1.json 2> $O/tier.err
python bench.py --bf16-tier --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline > $O/${R}_bench_bf16_tier_b8.json 2>> $O/tier.err
ls -la $O | head -20
