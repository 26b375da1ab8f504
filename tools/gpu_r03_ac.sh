#!/bin/bash
# round 3: "unsloth:auto" -- bitwise test, and the operating point on an idle MI355X next to "unsloth" and off
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "selective_recompute" > $OUT/pytest_r03ac.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_r03ac.log
for gc in unsloth:auto off unsloth; do
  BENCH_GC=$gc timeout 400 python bench.py --steps 8 --warmup 3 --alt-steps 0 --no-cpu-baseline > $OUT/bench_r03ac_${gc/:/_}.json 2> $OUT/bench_r03ac.err || tail -5 $OUT/bench_r03ac.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_r03ac_${gc/:/_}.json").read().strip().splitlines()[-1])
print("gc=$gc:", d["value"], "tok/s", d["ms_per_step"], "ms", d["peak_vram_gb"], "GB", "gemm", d["roofline"]["frac"])
PY
done
echo "all done ($SECONDS s)"
