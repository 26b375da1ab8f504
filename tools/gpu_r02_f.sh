#!/bin/bash
# round 2, 6th GPU pass: mrope parity, rocprofv3 kernel stats of the bench at batch 4 and batch 1 (profiles/r02*)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_mrope.py tests/test_gpu_elementwise.py -q -m gpu > $OUT/pytest_mrope_r02f.log 2>&1
tail -12 $OUT/pytest_mrope_r02f.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_r02f -o bench -- python $R/bench.py --steps 3 --warmup 1 --alt-steps 0 --no-cpu-baseline > $OUT/prof_r02f.log 2>&1
tail -2 $OUT/prof_r02f.log | cut -c1-600
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_r02f_b1 -o bench -- python $R/bench.py --steps 3 --warmup 1 --alt-steps 0 --no-cpu-baseline --batch 1 > $OUT/prof_r02f_b1.log 2>&1
tail -2 $OUT/prof_r02f_b1.log | cut -c1-600
cd $R
for d in prof_r02f prof_r02f_b1; do
  DB=$(find $OUT/$d -name '*.db' | head -1)
  python tools/rocpd_stats.py $DB > $OUT/${d}_kernel_stats.csv 2>&1
  head -30 $OUT/${d}_kernel_stats.csv
  find $OUT/$d -size +8M -delete
done
