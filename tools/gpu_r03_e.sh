#!/bin/bash
# round 3, pass e: full fine-tuning (config 3) pieces, vision tests, log-prob gradient bisect
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python tools/debug/logprob_grad.py > $OUT/logprob_grad_r03e.txt 2>&1
cat $OUT/logprob_grad_r03e.txt | tail -12
timeout 1200 python -m pytest tests/test_gpu_full_finetune.py tests/test_vision.py tests/test_gpu_rl_drivers.py tests/test_gpu_elementwise.py tests/test_gpu_model.py -m gpu -q -x --durations=8 > $OUT/pytest_r03e.log 2>&1
echo "pytest rc=$?"; tail -60 $OUT/pytest_r03e.log
cat $OUT/config3_parity.json
