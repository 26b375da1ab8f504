#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python tools/debug/logprob_model_grad.py > $OUT/logprob_model_r03f.txt 2>&1
grep -v Unrecognized $OUT/logprob_model_r03f.txt | tail -60
timeout 900 python -m pytest tests/test_vision.py tests/test_gpu_rl_drivers.py -m gpu -q --durations=5 > $OUT/pytest_r03f.log 2>&1
echo "pytest rc=$?"; tail -40 $OUT/pytest_r03f.log
