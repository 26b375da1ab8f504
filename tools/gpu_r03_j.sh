#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_glu_fused.py -m gpu -q -x > $OUT/pytest_r03j.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_r03j.log
timeout 300 python tools/glu_fused_bench.py > $OUT/glu_fused_bench_r03j.jsonl 2>$OUT/glu_fused_bench_r03j.err
cat $OUT/glu_fused_bench_r03j.jsonl
