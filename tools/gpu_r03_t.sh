#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for V in "default|" "persist2|UAMD_GEMM_PERSIST=2" "persist0|UAMD_GEMM_PERSIST=0" "groupm4|UAMD_GEMM_GROUP_M=4" "groupm16|UAMD_GEMM_GROUP_M=16" "default2|"; do
  NAME=${V%%|*}; ENVV=${V#*|}
  env $ENVV timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(variant='$NAME', tokens_per_s=d['value'], ms_per_step=d['ms_per_step'], gemm_tflops=d['roofline']['achieved'])))"
done | tee $OUT/gemm_knobs_r03t.jsonl
