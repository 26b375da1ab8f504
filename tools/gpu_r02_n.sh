#!/bin/bash
# round 2, 14th GPU pass: forward attention with 64 q rows per wave, hidden AGPR accumulators (knob bit 0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_attention.py -q -m gpu -k "forward" > $OUT/pytest_attn64_r02n.log 2>&1
echo "rc=$?"; tail -15 $OUT/pytest_attn64_r02n.log
for V in 0 1 0 1; do
  echo "== UAMD_ATTN_VAR=$V"
  UAMD_ATTN_VAR=$V timeout 300 python tools/microbench.py --skip-gemm --tokens 8192 --out $OUT/microbench_attn_r02n_v$V.jsonl > $OUT/microbench_attn_r02n_v$V.log 2>&1
  grep attn_fwd $OUT/microbench_attn_r02n_v$V.jsonl
done
