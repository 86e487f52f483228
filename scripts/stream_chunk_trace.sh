#!/bin/bash
# One chunk of the streaming path (BASELINE config 5) launch by launch: everything between two launches of the persistent sampler.
#   gpurun -- bash scripts/stream_chunk_trace.sh
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=gpurun_out/stream_chunk; mkdir -p $O
python scripts/time_stream.py --chunks 10 2>/dev/null | grep workload > $O/time_stream.json; cat $O/time_stream.json
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python $GRAFT_REPO_ROOT/scripts/time_stream.py --chunks 4 --no-parts > $GRAFT_REPO_ROOT/$O/tr.log 2>&1)
python scripts/trace_reduce.py $O/tr --end stream_step_kernel --rows > $O/chunk_trace.jsonl
rm -rf $O/tr
head -c 1500 $O/chunk_trace.jsonl
