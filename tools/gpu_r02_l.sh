#!/bin/bash
# round 2, 12th GPU pass: decode path tests, decode bench, per-kernel profile of the decode step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decode.py -q -m gpu -x > $OUT/pytest_decode_r02l.log 2>&1
echo "rc=$?"; tail -25 $OUT/pytest_decode_r02l.log
timeout 600 python tools/decode_bench.py --layers 32 --context 2048 --new 64 --out $OUT/decode_r02l.jsonl > $OUT/decode_r02l.log 2>&1
echo "rc=$?"; tail -9 $OUT/decode_r02l.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_decode_r02l -o decode -- python $R/tools/decode_bench.py --layers 8 --context 2048 --new 40 > $OUT/prof_decode_r02l.log 2>&1
cd $R
DB=$(find $OUT/prof_decode_r02l -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/decode_kernel_stats_r02l.csv 2>&1
rm -rf $OUT/prof_decode_r02l
head -30 $OUT/decode_kernel_stats_r02l.csv | cut -c1-200
