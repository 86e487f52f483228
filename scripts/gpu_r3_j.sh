#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
python scripts/time_codec.py --rounds 20 2>/dev/null | grep workload | tee $O/j_codec.jsonl
python scripts/time_encoders.py 2>/dev/null | grep workload | tee -a $O/j_codec.jsonl
for b in 1 8; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python $GRAFT_REPO_ROOT/scripts/time_codec.py --rounds 3 --batches $b --only encode > $GRAFT_REPO_ROOT/$O/tr.log 2>&1)
  python scripts/trace_reduce.py $O/tr --end pqmf_forward --rows > $O/j_encode_trace_b$b.jsonl
  rm -rf $O/tr
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python $GRAFT_REPO_ROOT/scripts/time_codec.py --rounds 3 --batches $b --only decode > $GRAFT_REPO_ROOT/$O/tr.log 2>&1)
  python scripts/trace_reduce.py $O/tr --end pqmf_inverse --rows > $O/j_decode_trace_b$b.jsonl
  rm -rf $O/tr
done
python bench.py --pmc > $O/j_pmc_b1.log 2>&1; tail -2 $O/j_pmc_b1.log
python bench.py --pmc --batch-per-gpu 8 > $O/j_pmc_b8.log 2>&1; tail -2 $O/j_pmc_b8.log
cp profiles/r3_pmc_hbm_*.json $O/ 2>/dev/null
ls $O | grep j_
