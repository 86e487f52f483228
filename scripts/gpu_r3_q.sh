#!/bin/bash
# full GPU suite + smoke after the persistent streaming sampler
mkdir -p gpurun_out/q
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/q/t_gpu.log
timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -3 > gpurun_out/q/smoke.log
cat gpurun_out/q/*.log
