#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_baseline_fullsize.py -m gpu -q -k config5 > $OUT/pytest_r03g1.log 2>&1
echo "alone rc=$?"; tail -5 $OUT/pytest_r03g1.log; python -c "
import json; d=json.load(open('gpurun_out/fullsize_parity.json')); print(json.dumps(d.get('config5_logprobs'), indent=1))"
timeout 900 python -m pytest tests/test_gpu_baseline_fullsize.py -m gpu -q > $OUT/pytest_r03g2.log 2>&1
echo "file rc=$?"; tail -5 $OUT/pytest_r03g2.log; python -c "
import json; d=json.load(open('gpurun_out/fullsize_parity.json')); print(json.dumps(d.get('config5_logprobs'), indent=1))"
