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
# same-box A/B of the two conditioning encoders side by side (default) against one after the other
for m in 0 1 0 1; do
  AFTER_ENCODERS_CONCURRENT=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'b1', 'AFTER_ENCODERS_CONCURRENT': $m, 'ms_per_step': d['ms_per_step'], 'xrt': d['value']}))" >> $O/${R}_ab_encoders_concurrent.jsonl
  AFTER_ENCODERS_CONCURRENT=$m python bench.py --steps 6 --warmup 2 --batch-per-gpu 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'leg': 'b8', 'AFTER_ENCODERS_CONCURRENT': $m, 'ms_per_step': d['ms_per_step'], 'xrt': d['value']}))" >> $O/${R}_ab_encoders_concurrent.jsonl
done
cat $O/${R}_ab_encoders_concurrent.jsonl
