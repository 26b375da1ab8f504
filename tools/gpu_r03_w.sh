#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_finetune.py -m gpu -q -k "return_logits or biased or every_gradient" > $OUT/pytest_r03w.log 2>&1
echo "pytest rc=$?"; grep -v "^$" $OUT/pytest_r03w.log | tail -30
