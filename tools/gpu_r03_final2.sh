#!/bin/bash
# round 3, second session, final pass: whole -m gpu suite, smoke, the driver's default bench command, kernel stats + last step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_r03final2.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -12 $OUT/pytest_r03final2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_r03final2.json 2> $OUT/bench_r03final2.err
echo "bench rc=$? ($SECONDS s)"; python - <<PY
import json
d=json.loads(open("$OUT/bench_r03final2.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","peak_vram_gb","steps","warmup","loss_first_last")}, d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["steps_sampled"], d["cpu_baseline"]["value"])
for k,v in d["alt"].items(): print(" ", k[:60], {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","peak_vram_gb","tokens_per_s","ms_per_token","gemm_frac_of_mfma_peak","frac_of_hbm_peak","error")})
PY
tail -2 $OUT/bench_r03final2.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r03final2 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03final2.log 2>&1
cd $R
DB=$(find $OUT/prof_r03final2 -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r03final2_bench_kernel_stats.csv 2> $OUT/r03final2_stats.err
python tools/rocpd_sequence.py $DB > $OUT/r03final2_step_sequence.csv 2> $OUT/r03final2_seq.err
head -12 $OUT/r03final2_bench_kernel_stats.csv | cut -c1-140; grep "^# kernels" $OUT/r03final2_step_sequence.csv
rm -rf $OUT/prof_r03final2
echo "all done ($SECONDS s)"
