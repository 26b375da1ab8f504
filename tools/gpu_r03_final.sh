#!/bin/bash
# round 3, final pass: whole -m gpu suite, smoke, the driver's bench command, kernel stats + last step in launch order
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_r03final.log 2>&1
echo "pytest rc=$?"; tail -16 $OUT/pytest_r03final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r03final.json 2> $OUT/bench_r03final.err
cat $OUT/bench_r03final.json; tail -3 $OUT/bench_r03final.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r03final -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03final.log 2>&1
cd $R
DB=$(find $OUT/prof_r03final -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r03final_bench_kernel_stats.csv 2> $OUT/r03final_stats.err
python tools/rocpd_sequence.py $DB > $OUT/r03final_step_sequence.csv 2> $OUT/r03final_seq.err
head -14 $OUT/r03final_bench_kernel_stats.csv; grep "^#" $OUT/r03final_step_sequence.csv | head -3
rm -rf $OUT/prof_r03final
