#!/bin/bash
# round 3, pass k: whole -m gpu suite, smoke, default bench (with the config-3 alt point), batch-1 kernel sequence
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_r03k.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest_r03k.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > $OUT/bench_r03k.json 2> $OUT/bench_r03k.err
cat $OUT/bench_r03k.json; tail -3 $OUT/bench_r03k.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r03k -o bench -- python $R/bench.py --batch 1 --steps 3 --warmup 2 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03k.log 2>&1
cd $R
DB=$(find $OUT/prof_r03k -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r03k_batch1_kernel_stats.csv 2> $OUT/r03k_stats.err
python tools/rocpd_sequence.py $DB > $OUT/r03k_batch1_step_sequence.csv 2> $OUT/r03k_seq.err
head -24 $OUT/r03k_batch1_kernel_stats.csv; grep "^#" $OUT/r03k_batch1_step_sequence.csv | head -8
rm -rf $OUT/prof_r03k
