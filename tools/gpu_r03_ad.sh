#!/bin/bash
# round 3: persistent GEMM without the compiler's vmcnt(0) in its K loop (PLAIN kernel instance): parity, microbench, step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_nf4_gemm.py -m gpu -q -k "gemm256" > $OUT/pytest_r03ad.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -3 $OUT/pytest_r03ad.log
timeout 400 python tools/gemm_plain_ab.py > $OUT/r03ad_gemm_plain_ab.jsonl 2> $OUT/gemm_plain_ab.err
echo "gemm ab rc=$? ($SECONDS s)"; cat $OUT/r03ad_gemm_plain_ab.jsonl; tail -3 $OUT/gemm_plain_ab.err
B="python bench.py --gpus 1 --steps 12 --warmup 4 --alt-steps 0 --no-cpu-baseline"
for cfg in "0 1" "1 1" "1 2" "0 1" "1 1" "1 2"; do
  set -- $cfg
  UAMD_GEMM_PLAIN=$1 UAMD_GEMM_PERSIST=$2 timeout 400 $B > $OUT/bench_r03ad_plain$1_persist$2.json 2> $OUT/bench_r03ad.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_r03ad_plain$1_persist$2.json").read().strip().splitlines()[-1])
print("PLAIN=$1 PERSIST=$2:", d["value"], "tok/s", d["ms_per_step"], "ms", "gemm", d["roofline"]["achieved"], d["roofline"]["frac"])
PY
done
echo "all done ($SECONDS s)"
