#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel stats. Outputs under gpurun_out/.
# usage (from the repo root): gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1
  tail -3 $OUT/pytest_gpu_$TAG.log
fi
timeout 900 python bench.py ${BENCH_ARGS} > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
cat $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
if [ -z "$SKIP_PROF" ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $OUT/prof_$TAG.log 2>&1
  tail -2 $OUT/prof_$TAG.log
  find $OUT/prof_$TAG -name '*kernel_stats*' | head
  # keep only the summaries (traces are large)
  find $OUT/prof_$TAG -name '*kernel_trace*' -size +8M -delete
fi
