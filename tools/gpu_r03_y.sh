#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api_surface.py -m gpu -q > $OUT/pytest_r03y.log 2>&1
echo "pytest rc=$?"; grep -v "^$" $OUT/pytest_r03y.log | tail -60
