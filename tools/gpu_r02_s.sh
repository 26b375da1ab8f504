#!/bin/bash
# round 2, pass s: full GPU regression + smoke on the current code
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0; timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu_r02s.log 2>&1
echo "pytest rc=$?"; tail -30 $OUT/pytest_gpu_r02s.log; echo "pytest seconds: $SECONDS"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
