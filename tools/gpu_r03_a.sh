#!/bin/bash
# round 3, pass a: full GPU regression (no -x: measured values of the tightened bounds) + smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0; timeout 1700 python -m pytest tests -m gpu -q --durations=20 > $OUT/pytest_gpu_r03a.log 2>&1
echo "pytest rc=$?"; tail -60 $OUT/pytest_gpu_r03a.log; echo "pytest seconds: $SECONDS"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cat $OUT/fullsize_parity.json 2>/dev/null | head -80
