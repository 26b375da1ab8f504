#!/bin/bash
# round 2, 9th GPU pass: persistent tile walk of the 256x256 GEMM (guarded), full regression, GEMM A/B, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_nf4_gemm.py -q -m gpu -x -k "persistent" > $OUT/pytest_persist_r02i.log 2>&1
RC=$?
tail -12 $OUT/pytest_persist_r02i.log
if [ $RC -ne 0 ]; then echo "PERSISTENT GEMM FAILED rc=$RC: one block per tile for the rest of this pass"; export UAMD_GEMM_PERSIST=0; fi
timeout 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_nf4_gemm.py::test_gemm256_persistent_walk_is_bit_identical > $OUT/pytest_gpu_r02i.log 2>&1
tail -12 $OUT/pytest_gpu_r02i.log
if [ $RC -eq 0 ]; then
  for P in 1 0 1 0; do
    UAMD_GEMM_PERSIST=$P timeout 300 python tools/microbench.py --only-gemm --gemm-tokens 8192 --out $OUT/microbench_gemm_r02i_p$P.jsonl > $OUT/microbench_gemm_r02i_p$P.log 2>&1
    echo "== persist=$P"; python - <<PY
import json
for l in open("$OUT/microbench_gemm_r02i_p$P.jsonl"):
    try: d=json.loads(l)
    except Exception: continue
    if "gemm" in d.get("kernel","") or "matmul" in d.get("kernel",""): print(d.get("kernel")[:60], d.get("us"), d.get("TFLOPs"))
PY
  done
fi
timeout 900 python bench.py --steps 8 --warmup 2 --alt-steps 3 --no-cpu-baseline > $OUT/bench_r02i.json 2> $OUT/bench_r02i.err
cat $OUT/bench_r02i.json; tail -5 $OUT/bench_r02i.err
