#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -x > $OUT/pytest_r03n.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_r03n.log
timeout 200 python tools/attn_bwd_ab.py > $OUT/attn_bwd_ab_r03n.jsonl 2>$OUT/attn_bwd_ab_r03n.err
cat $OUT/attn_bwd_ab_r03n.jsonl; tail -2 $OUT/attn_bwd_ab_r03n.err
ATTN_SHAPE=1,8,2,512,128 timeout 200 python tools/attn_bwd_ab.py 2>/dev/null | tail -4
