#!/bin/bash
# round 2, pass p: decode-ahead (side-stream NF4 decode) -- bitwise tests, key-padding / padded-head attention tests, bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decode_ahead.py tests/test_gpu_model.py -m gpu -x -q > $OUT/pytest_r02p.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_r02p.log
timeout 900 python bench.py > $OUT/bench_r02p.json 2> $OUT/bench_r02p.err
echo "bench rc=$?"; tail -3 $OUT/bench_r02p.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_r02p.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","peak_vram_gb","steps","warmup")}, d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
for k,v in d["alt"].items(): print(" ", k[:70], {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","peak_vram_gb","tokens_per_s","ms_per_token","gemm_tflops")})
PY
