#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q -x > $OUT/pytest_r03s.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_r03s.log
for F in 0 1; do
  UNSLOTH_AMD_DECODE_FUSED=$F timeout 300 python tools/decode_bench.py 2>/dev/null | tail -7
done | tee $OUT/decode_ab_r03s.jsonl
