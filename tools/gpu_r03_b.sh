#!/bin/bash
# round 3, pass b: attention tests with the new dK/dV kernel, A/B timing, then the full regression
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_attention.py -q -x > $OUT/pytest_attn_r03b.log 2>&1
echo "attention pytest rc=$?"; tail -15 $OUT/pytest_attn_r03b.log
timeout 300 python tools/attn_bwd_ab.py > $OUT/attn_bwd_ab_r03b.jsonl 2> $OUT/attn_bwd_ab_r03b.err
cat $OUT/attn_bwd_ab_r03b.jsonl; tail -3 $OUT/attn_bwd_ab_r03b.err
ATTN_SHAPE=1,32,8,512,128 timeout 300 python tools/attn_bwd_ab.py > $OUT/attn_bwd_ab_small_r03b.jsonl 2>&1; cat $OUT/attn_bwd_ab_small_r03b.jsonl
SECONDS=0; timeout 1700 python -m pytest tests -m gpu -q --deselect tests/test_gpu_attention.py > $OUT/pytest_gpu_r03b.log 2>&1
echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu_r03b.log; echo "pytest seconds: $SECONDS"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cat $OUT/fullsize_parity.json 2>/dev/null
