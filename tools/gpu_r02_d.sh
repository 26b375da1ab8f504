#!/bin/bash
# round 2, 4th GPU pass: regression (selective-recompute layer Function, SGPR-base GEMM DMA), GEMM microbench, bench with alts
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp $OUT/ref_triton_bf16.pt tests/golden/ 2>/dev/null
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02d.log 2>&1
tail -25 $OUT/pytest_gpu_r02d.log
timeout 400 python tools/microbench.py --only-gemm --gemm-tokens 8192 --out $OUT/microbench_gemm_r02d.jsonl > $OUT/microbench_gemm_r02d.log 2>&1
grep -i "gemm\|matmul\|error" $OUT/microbench_gemm_r02d.jsonl
timeout 900 python bench.py --steps 8 --warmup 2 --alt-steps 3 --no-cpu-baseline > $OUT/bench_r02d.json 2> $OUT/bench_r02d.err
cat $OUT/bench_r02d.json; tail -5 $OUT/bench_r02d.err
