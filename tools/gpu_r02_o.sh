#!/bin/bash
# round 2, 15th GPU pass: VALU-trimmed attention kernels (fma exponent, lazy rescale, packed conversions): tests + timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_model.py -q -m gpu -x > $OUT/pytest_attn_r02o.log 2>&1
echo "rc=$?"; tail -5 $OUT/pytest_attn_r02o.log
for V in 0 1 0; do
  UAMD_ATTN_VAR=$V timeout 300 python tools/microbench.py --skip-gemm --tokens 8192 2>/dev/null | grep attn
done
