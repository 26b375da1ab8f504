#!/bin/bash
# round 2, 10th GPU pass: forward attention with 64 q rows per wave (experiment), regression of the rest
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_attention.py -q -m gpu -k "forward_matches" > $OUT/pytest_attn64_r02j.log 2>&1
tail -15 $OUT/pytest_attn64_r02j.log
for V in 0 1 0 1; do
  echo "== UAMD_ATTN_VAR=$V"
  UAMD_ATTN_VAR=$V timeout 300 python tools/microbench.py --skip-gemm --tokens 8192 --out $OUT/microbench_attn_r02j_v$V.jsonl > $OUT/microbench_attn_r02j_v$V.log 2>&1
  grep attn $OUT/microbench_attn_r02j_v$V.jsonl
done
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02j.log 2>&1
tail -6 $OUT/pytest_gpu_r02j.log
