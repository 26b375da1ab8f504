#!/bin/bash
# round 3: attention forward / dQ without the compiler's vmcnt countdown on the resident Q / dO fragments: parity, kernel
# A/B against the previous attention.hip (same box, library built beside: lib/libunsloth_amd_prevattn.so), whole step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q -x > $OUT/pytest_r03ae.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -3 $OUT/pytest_r03ae.log
# (build it first: the objects of unsloth_amd/lib/*.o with attention.o replaced by the previous attention.hip, `hipcc -shared`)
PREV=$R/unsloth_amd/lib/libunsloth_amd_prevattn.so
for i in 1 2; do
  echo "prev:"; UNSLOTH_AMD_LIB=$PREV timeout 200 python tools/attn_bench.py 2>/dev/null | tee -a $OUT/r03ae_attn_ab_prev.jsonl
  echo "new:"; timeout 200 python tools/attn_bench.py 2>/dev/null | tee -a $OUT/r03ae_attn_ab_new.jsonl
done
B="python bench.py --gpus 1 --steps 12 --warmup 4 --alt-steps 0 --no-cpu-baseline"
for v in prev new prev new; do
  if [ $v = prev ]; then export UNSLOTH_AMD_LIB=$PREV; else unset UNSLOTH_AMD_LIB; fi
  timeout 400 $B > $OUT/bench_r03ae_$v.json 2> $OUT/bench_r03ae.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_r03ae_$v.json").read().strip().splitlines()[-1])
print("attention $v:", d["value"], "tok/s", d["ms_per_step"], "ms", "gemm", d["roofline"]["achieved"], d["roofline"]["frac"])
PY
done
unset UNSLOTH_AMD_LIB
echo "all done ($SECONDS s)"
