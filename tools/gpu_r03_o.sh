#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for KO in 0 1 2 4 8 3 7; do
  UAMD_ATTN_CFLAGS="-fno-slp-vectorize -DUAMD_DQ4_KO=$KO" python -c "
import os
from unsloth_amd import _build
os.remove(os.path.join(_build.LIBDIR, 'attention.o'))
_build.build()" > $OUT/build_ko$KO.log 2>&1 || { echo "build $KO failed"; tail -3 $OUT/build_ko$KO.log; continue; }
  TAG="ko=$KO" timeout 120 python tools/attn_dq_only.py 2>/dev/null | tail -1
done | tee $OUT/dq4_knockout_r03o.jsonl
python -c "
import os
from unsloth_amd import _build
os.remove(os.path.join(_build.LIBDIR, 'attention.o'))
_build.build()"
