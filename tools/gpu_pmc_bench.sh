#!/bin/bash
# HBM-side traffic of the step's kernels: two separate PMC passes over a short bench run (FETCH_SIZE needs 3 TCC
# slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), per-kernel mean per dispatch.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_pmc_bench.sh tag'
TAG=${1:-pmcb}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $SET -d $OUT/${TAG}_$i -o pmc -- python $R/bench.py --steps 2 --warmup 1 --alt-steps 0 --no-cpu-baseline > $OUT/${TAG}_$i.log 2>&1
  tail -1 $OUT/${TAG}_$i.log | cut -c1-200
  DB=$(find $OUT/${TAG}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB > $OUT/${TAG}_$i.txt 2>&1 && grep -A1 "gemm_nt256\|attn_\|nf4_dequant\|glu_" $OUT/${TAG}_$i.txt | head -40
  [ -n "$DB" ] && rm -f $DB
done
