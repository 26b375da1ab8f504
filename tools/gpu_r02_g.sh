#!/bin/bash
# round 2, 7th GPU pass: NN GEMM form (guarded), class-level patch test, full regression, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_nf4_gemm.py -q -m gpu -x -k "nn_form" > $OUT/pytest_nn_r02g.log 2>&1
RC=$?
tail -12 $OUT/pytest_nn_r02g.log
if [ $RC -ne 0 ]; then echo "NN GEMM FAILED rc=$RC: falling back to the transposed-decode path for the rest of this pass"; export UNSLOTH_AMD_NN_DX=0; fi
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_nf4_gemm.py::test_gemm256_nn_form > $OUT/pytest_gpu_r02g.log 2>&1
tail -12 $OUT/pytest_gpu_r02g.log
timeout 900 python bench.py --steps 8 --warmup 2 --alt-steps 3 --no-cpu-baseline > $OUT/bench_r02g.json 2> $OUT/bench_r02g.err
cat $OUT/bench_r02g.json; tail -5 $OUT/bench_r02g.err
