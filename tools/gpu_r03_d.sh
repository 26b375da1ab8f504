#!/bin/bash
# round 3, pass d: whole -m gpu suite (vision + RL drivers new), smoke, bench, kernel stats from the rocpd database
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_r03d.log 2>&1
echo "pytest rc=$?"; tail -40 $OUT/pytest_r03d.log
cat $OUT/fullsize_parity.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > $OUT/bench_r03d.json 2> $OUT/bench_r03d.err
cat $OUT/bench_r03d.json; tail -3 $OUT/bench_r03d.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r03d -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03d.log 2>&1
cd $R
DB=$(find $OUT/prof_r03d -name '*.db' | head -1)
echo "db=$DB"
python tools/rocpd_stats.py $DB > $OUT/r03d_bench_kernel_stats.csv 2> $OUT/r03d_stats.err
python tools/rocpd_sequence.py $DB > $OUT/r03d_step_sequence.csv 2> $OUT/r03d_step_sequence.err
head -30 $OUT/r03d_bench_kernel_stats.csv
rm -rf $OUT/prof_r03d
