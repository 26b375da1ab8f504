#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp
for V in "default|" "half_always|UAMD_GEMM_HALF=2" "half_never|UAMD_GEMM_HALF=0"; do
  NAME=${V%%|*}; ENVV=${V#*|}
  env $ENVV timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(variant='$NAME', tokens_per_s=d['value'], ms_per_step=d['ms_per_step'], gemm_tflops=d['roofline']['achieved'])))"
done | tee $OUT/gemm_half_r03u.jsonl
env UAMD_GEMM_HALF=2 python tools/gemm_ab.py $OUT/gemm_ab_half_r03u.jsonl > /dev/null 2>&1; cat $OUT/gemm_ab_half_r03u.jsonl
python tools/gemm_ab.py $OUT/gemm_ab_def_r03u.jsonl > /dev/null 2>&1; cat $OUT/gemm_ab_def_r03u.jsonl
