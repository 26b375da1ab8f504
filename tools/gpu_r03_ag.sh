#!/bin/bash
# round 3: the full-size BASELINE configurations, full fine-tuning, decode and RL drivers at HEAD (PLAIN persistent GEMM)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 330 python -m pytest tests/test_gpu_baseline_fullsize.py tests/test_gpu_full_finetune.py tests/test_gpu_rl_drivers.py tests/test_gpu_dp_rccl.py -m gpu -q -x > $OUT/pytest_r03ag.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_r03ag.log
echo "all done ($SECONDS s)"
