#!/bin/bash
# kernel census of ONE step at batch 1 (T = 2048 tokens: SURVEY's config-2 default)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT/prof_bs1 -o bench -- python $R/bench.py --batch 1 --steps 2 --warmup 1 --alt-steps 0 --no-cpu-baseline > $OUT/prof_bs1.log 2>&1
cd $R
DB=$(find $OUT/prof_bs1 -name '*.db' | head -1)
python tools/rocpd_sequence.py $DB > $OUT/r02zz_step_sequence_batch1.csv 2>/dev/null
rm -rf $OUT/prof_bs1
grep "^#" $OUT/r02zz_step_sequence_batch1.csv | head -4 | cut -c1-200
awk -F, 'NR>1 && $1!~/^#/ {n[$4]++; t[$4]+=$2} END{for(k in n) printf "%8.2f ms %5d  %s\n", t[k]/1000, n[k], k}' $OUT/r02zz_step_sequence_batch1.csv | sort -rn | head -24
