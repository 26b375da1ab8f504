#!/bin/bash
# round 3, re-entry pass at HEAD: whole -m gpu suite, smoke, the primary bench point, kernel stats + last step in launch
# order, then the PMC campaign (tools/gpu_r03_pmc.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $OUT/pytest_r03z.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -14 $OUT/pytest_r03z.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 4 --alt-steps 0 --no-cpu-baseline > $OUT/bench_r03z.json 2> $OUT/bench_r03z.err
echo "bench rc=$? ($SECONDS s)"; cut -c1-900 $OUT/bench_r03z.json; tail -2 $OUT/bench_r03z.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r03z -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03z.log 2>&1
cd $R
DB=$(find $OUT/prof_r03z -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r03z_bench_kernel_stats.csv 2> $OUT/r03z_stats.err
python tools/rocpd_sequence.py $DB > $OUT/r03z_step_sequence.csv 2> $OUT/r03z_seq.err
head -8 $OUT/r03z_bench_kernel_stats.csv | cut -c1-150; grep "^#" $OUT/r03z_step_sequence.csv | head -2
rm -rf $OUT/prof_r03z
echo "stats done ($SECONDS s)"
bash tools/gpu_r03_pmc.sh
echo "all done ($SECONDS s)"
