#!/bin/bash
# round 2, 8th GPU pass: full regression, bench with all alt points, PMC traffic passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02h.log 2>&1
tail -12 $OUT/pytest_gpu_r02h.log
timeout 900 python bench.py --steps 8 --warmup 2 --alt-steps 3 --no-cpu-baseline > $OUT/bench_r02h.json 2> $OUT/bench_r02h.err
cat $OUT/bench_r02h.json; tail -5 $OUT/bench_r02h.err
bash tools/gpu_pmc_bench.sh pmc_r02h > $OUT/pmc_r02h.log 2>&1
tail -30 $OUT/pmc_r02h.log
