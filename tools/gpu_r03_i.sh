#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_glu_fused.py -m gpu -q -x > $OUT/pytest_r03i.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest_r03i.log
timeout 300 python tools/glu_fused_bench.py > $OUT/glu_fused_bench_r03i.jsonl 2>$OUT/glu_fused_bench_r03i.err
cat $OUT/glu_fused_bench_r03i.jsonl; tail -3 $OUT/glu_fused_bench_r03i.err
for F in 1 0; do
  UNSLOTH_AMD_GLU_FUSED=$F timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --alt-steps 0 > $OUT/bench_r03i_fused$F.json 2> $OUT/bench_r03i_fused$F.err
  python -c "
import json; d=json.loads(open('$OUT/bench_r03i_fused$F.json').read().strip().splitlines()[-1]); print('fused=$F', d['value'], d['ms_per_step'])"
done
