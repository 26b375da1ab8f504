#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -x > $OUT/pytest_r03p.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_r03p.log
timeout 200 python tools/attn_bwd_ab.py > $OUT/attn_bwd_ab_r03p.jsonl 2>$OUT/attn_bwd_ab_r03p.err
cat $OUT/attn_bwd_ab_r03p.jsonl
for V in "pf2|-fno-slp-vectorize" "pf4|-fno-slp-vectorize -DUAMD_DQ4_PF=4" "pf3|-fno-slp-vectorize -DUAMD_DQ4_PF=3" "ko8|-fno-slp-vectorize -DUAMD_DQ4_KO=8" "ko1|-fno-slp-vectorize -DUAMD_DQ4_KO=1" "ko4|-fno-slp-vectorize -DUAMD_DQ4_KO=4" "ko7|-fno-slp-vectorize -DUAMD_DQ4_KO=7"; do
  NAME=${V%%|*}; FLAGS=${V#*|}
  UAMD_ATTN_CFLAGS="$FLAGS" python -c "
import os
from unsloth_amd import _build
os.remove(os.path.join(_build.LIBDIR, 'attention.o'))
_build.build()" > $OUT/build_$NAME.log 2>&1 || { echo "build $NAME failed"; tail -3 $OUT/build_$NAME.log; continue; }
  TAG="$NAME" timeout 120 python tools/attn_dq_only.py 2>/dev/null | tail -1
done | tee $OUT/dq4_variants_r03p.jsonl
python -c "
import os
from unsloth_amd import _build
os.remove(os.path.join(_build.LIBDIR, 'attention.o'))
_build.build()"
