#!/bin/bash
# round 2, FINAL pass (second session): full regression, the driver's default bench command, kernel stats, HBM traffic PMC
# passes (pmc_traffic.json), MFMA-busy / effective-clock PMC pass
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_r02zz.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -3 $OUT/pytest_gpu_r02zz.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > $OUT/bench_r02zz.json 2> $OUT/bench_r02zz.err
echo "bench rc=$? ($SECONDS s)"; python - <<PY
import json
d=json.loads(open("$OUT/bench_r02zz.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","peak_vram_gb","steps","warmup")}, d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["config"]["optimizer"])
for k,v in d["alt"].items(): print(" ", k[:70], {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","peak_vram_gb","tokens_per_s","ms_per_token")})
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_r02zz -o bench -- python $R/bench.py --steps 3 --warmup 1 --alt-steps 0 --no-cpu-baseline > $OUT/prof_r02zz.log 2>&1
cd $R
DB=$(find $OUT/prof_r02zz -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r02zz_bench_kernel_stats.csv 2>&1
python tools/rocpd_sequence.py $DB > $OUT/r02zz_step_sequence.csv 2>/dev/null
head -12 $OUT/r02zz_bench_kernel_stats.csv | cut -c1-150; grep "^# kernels" $OUT/r02zz_step_sequence.csv
rm -rf $OUT/prof_r02zz
echo "stats done ($SECONDS s)"
bash tools/gpu_pmc_bench.sh pmc_r02zz > $OUT/pmc_r02zz.log 2>&1
tail -4 $OUT/pmc_r02zz.log | cut -c1-160
echo "pmc traffic done ($SECONDS s)"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc_r02zz_mfma -o pmc -- python $R/bench.py --steps 2 --warmup 1 --alt-steps 0 --no-cpu-baseline > $OUT/pmc_r02zz_mfma.log 2>&1
cd $R
DB=$(find $OUT/pmc_r02zz_mfma -name '*.db' | head -1)
[ -n "$DB" ] && python tools/pmc_summary.py $DB > $OUT/pmc_r02zz_mfma.txt 2>&1 && rm -rf $OUT/pmc_r02zz_mfma
grep -A7 "^gemm_nt256\|^attn_" $OUT/pmc_r02zz_mfma.txt | head -70
echo "all done ($SECONDS s)"
