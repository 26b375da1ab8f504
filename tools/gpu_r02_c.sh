#!/bin/bash
# round 2, third GPU pass: final (small) bf16 golden, regression with the new defaults (row-per-block RMSNorm,
# two-vector GLU, peeled rank-block GEMM loop), HBM A/B incl. row-fastest transposed dequant, GEMM microbench, bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python oracle/make_golden_bf16_gpu.py --out $OUT/ref_triton_bf16.pt > $OUT/golden_bf16.log 2>&1
tail -3 $OUT/golden_bf16.log
cp $OUT/ref_triton_bf16.pt tests/golden/ 2>/dev/null
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02c.log 2>&1
tail -15 $OUT/pytest_gpu_r02c.log
HBM_AB_ONLY=dequant timeout 400 python tools/hbm_ab.py $OUT/hbm_ab_r02c.jsonl > $OUT/hbm_ab_r02c.log 2>&1
cat $OUT/hbm_ab_r02c.jsonl
timeout 400 python tools/microbench.py --only-gemm --gemm-tokens 8192 --out $OUT/microbench_gemm_r02c.jsonl > $OUT/microbench_gemm_r02c.log 2>&1
grep -i "gemm\|matmul\|error" $OUT/microbench_gemm_r02c.jsonl
timeout 600 python bench.py --steps 6 --warmup 2 --alt-steps 0 --no-cpu-baseline > $OUT/bench_r02c.json 2> $OUT/bench_r02c.err
cat $OUT/bench_r02c.json; tail -3 $OUT/bench_r02c.err
