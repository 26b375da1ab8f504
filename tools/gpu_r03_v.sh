#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vision.py tests/test_gpu_baseline_fullsize.py -m gpu -q -k "vision or pixel" > $OUT/pytest_r03v.log 2>&1
echo "pytest rc=$?"; grep -v "^$" $OUT/pytest_r03v.log | tail -30
python -c "
import json; d=json.load(open('gpurun_out/fullsize_parity.json')); print(json.dumps(d.get('config4_pixels_real_widths'), indent=1))"
timeout 300 python -m pytest tests/test_gpu_dp_rccl.py -m gpu -q -k full_finetune 2>&1 | tail -3
