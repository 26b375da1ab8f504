#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_vision.py -m gpu -q > $OUT/pytest_r03x.log 2>&1
echo "pytest rc=$?"; grep -v "^$" $OUT/pytest_r03x.log | tail -40
