#!/bin/bash
# round 2, 13th GPU pass: full regression (incl. decode path), bench with the decode alt point
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02m.log 2>&1
echo "rc=$?"; tail -8 $OUT/pytest_gpu_r02m.log
timeout 900 python bench.py --steps 8 --warmup 2 --alt-steps 3 --no-cpu-baseline > $OUT/bench_r02m.json 2> $OUT/bench_r02m.err
echo "rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/bench_r02m.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","peak_vram_gb")}, d["roofline"]["achieved"], d["roofline"]["frac"])
for k,v in d["alt"].items(): print(" ", k[:60], {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","peak_vram_gb","tokens_per_s","ms_per_token","frac_of_hbm_peak")})
PY
tail -3 $OUT/bench_r02m.err
