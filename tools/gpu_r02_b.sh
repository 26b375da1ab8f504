#!/bin/bash
# round 2, second GPU pass: smaller bf16 golden, parity, full regression on the rank-block-K-tile GEMM, HBM kernel A/B,
# GEMM microbench, short bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python oracle/make_golden_bf16_gpu.py --out $OUT/ref_triton_bf16.pt > $OUT/golden_bf16.log 2>&1
tail -3 $OUT/golden_bf16.log
cp $OUT/ref_triton_bf16.pt tests/golden/ 2>/dev/null
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02b.log 2>&1
tail -15 $OUT/pytest_gpu_r02b.log
timeout 400 python tools/hbm_ab.py $OUT/hbm_ab_r02b.jsonl > $OUT/hbm_ab_r02b.log 2>&1
cat $OUT/hbm_ab_r02b.jsonl
timeout 400 python tools/microbench.py --only-gemm --gemm-tokens 8192 2048 --out $OUT/microbench_gemm_r02b.jsonl > $OUT/microbench_gemm_r02b.log 2>&1
cat $OUT/microbench_gemm_r02b.jsonl
timeout 600 python bench.py --steps 6 --warmup 2 --alt-steps 0 --no-cpu-baseline > $OUT/bench_r02b.json 2> $OUT/bench_r02b.err
cat $OUT/bench_r02b.json; tail -3 $OUT/bench_r02b.err
