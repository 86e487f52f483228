#!/bin/bash
# where a streaming chunk's time goes: kernel trace of bench.py --stream (persistent step)
mkdir -p gpurun_out/o
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o -- python $GRAFT_REPO_ROOT/bench.py --stream --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_o.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_o -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/o/r3_stream_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over 8 chunks (2 warm-up + 6 timed), {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:70]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms {int(r['Calls']):7d} calls {float(r['AverageNs'])/1e3:9.2f} us avg  {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
PY
grep '^{' /tmp/prof_o.log | cut -c1-200 >> gpurun_out/o/r3_stream_kernel_stats.txt
cat gpurun_out/o/r3_stream_kernel_stats.txt
