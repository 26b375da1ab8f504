#!/bin/bash
# round 2, pass q: padded attention shapes (G = 3/5/6/7, head_dim < 128), key-padding documents, decode attention G = 1..8, error probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_mrope.py tests/test_gpu_model.py -m gpu -q > $OUT/pytest_r02q.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest_r02q.log
timeout 300 python -m pytest tests/test_gpu_decode.py -m gpu -q -k "attn_decode" > $OUT/pytest_r02q_decode.log 2>&1
echo "pytest decode rc=$?"; tail -4 $OUT/pytest_r02q_decode.log
timeout 300 python tools/attn_err_probe.py > $OUT/attn_err_r02q.jsonl 2> $OUT/attn_err_r02q.err
echo "probe rc=$?"; cat $OUT/attn_err_r02q.jsonl; tail -3 $OUT/attn_err_r02q.err
