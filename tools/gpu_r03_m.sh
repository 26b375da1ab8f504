#!/bin/bash
# round 3, pass m: bench + kernel stats with the fixed GEMM epilogue
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_nf4_gemm.py tests/test_gpu_lora_blocks.py tests/test_gpu_full_finetune.py -m gpu -q > $OUT/pytest_r03m.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_r03m.log
timeout 1200 python bench.py --steps 12 --warmup 3 > $OUT/bench_r03m.json 2> $OUT/bench_r03m.err
cat $OUT/bench_r03m.json; tail -3 $OUT/bench_r03m.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r03m -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03m.log 2>&1
cd $R
DB=$(find $OUT/prof_r03m -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r03m_bench_kernel_stats.csv 2> $OUT/r03m_stats.err
python tools/rocpd_sequence.py $DB > $OUT/r03m_step_sequence.csv 2> $OUT/r03m_seq.err
head -16 $OUT/r03m_bench_kernel_stats.csv; grep "^#" $OUT/r03m_step_sequence.csv | head -3
rm -rf $OUT/prof_r03m
