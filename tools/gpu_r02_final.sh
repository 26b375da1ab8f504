#!/bin/bash
# round 2, final GPU pass: full regression, the driver's default bench command, kernel stats + PMC traffic of the final code
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02z.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu_r02z.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > $OUT/bench_r02z.json 2> $OUT/bench_r02z.err
echo "bench rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/bench_r02z.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","peak_vram_gb","steps","warmup")}, d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"])
for k,v in d["alt"].items(): print(" ", k[:70], {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","peak_vram_gb","tokens_per_s","ms_per_token")})
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_r02z -o bench -- python $R/bench.py --steps 3 --warmup 1 --alt-steps 0 --no-cpu-baseline > $OUT/prof_r02z.log 2>&1
cd $R
DB=$(find $OUT/prof_r02z -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r02z_bench_kernel_stats.csv 2>&1
head -14 $OUT/r02z_bench_kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof_r02z
bash tools/gpu_pmc_bench.sh pmc_r02z > $OUT/pmc_r02z.log 2>&1
tail -6 $OUT/pmc_r02z.log | cut -c1-200
