#!/bin/bash
# round 3, pass c: full-size config parity (fixed), LayerNorm, dK/dV kernel build variants A/B, bench + kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_baseline_fullsize.py tests/test_layernorm.py tests/test_gpu_nf4_gemm.py -m gpu -q > $OUT/pytest_r03c.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest_r03c.log
cat $OUT/fullsize_parity.json
# --- dK/dV kernel variants (attention.o only is rebuilt)
for V in "default|-fno-slp-vectorize" "dma8|-fno-slp-vectorize -DUAMD_KD4_DMA_CHUNK=8" "pf3|-fno-slp-vectorize -DUAMD_KD4_PF=3" "slp|-O3"; do
  NAME=${V%%|*}; FLAGS=${V#*|}
  UAMD_ATTN_CFLAGS="$FLAGS" python -c "
import os
from unsloth_amd import _build
os.remove(os.path.join(_build.LIBDIR, 'attention.o'))
_build.build()" > $OUT/build_$NAME.log 2>&1 || { echo "build $NAME failed"; tail -5 $OUT/build_$NAME.log; continue; }
  echo "== variant $NAME ($FLAGS)"
  timeout 200 python tools/attn_bwd_ab.py 2>/dev/null | head -3
done > $OUT/attn_variants_r03c.txt 2>&1
cat $OUT/attn_variants_r03c.txt
python -c "
import os
from unsloth_amd import _build
os.remove(os.path.join(_build.LIBDIR, 'attention.o'))
_build.build()"
# --- bench (short) + kernel stats
timeout 900 python bench.py --steps 8 --warmup 3 > $OUT/bench_r03c.json 2> $OUT/bench_r03c.err
cat $OUT/bench_r03c.json; tail -3 $OUT/bench_r03c.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03c -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03c.log 2>&1
tail -2 $OUT/prof_r03c.log
find $OUT/prof_r03c -name '*kernel_stats*' | head
find $OUT/prof_r03c -name '*kernel_trace*' -size +8M -delete
find $OUT/prof_r03c -name '*.db' -size +8M -delete
