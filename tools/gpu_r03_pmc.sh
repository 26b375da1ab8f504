#!/bin/bash
# round 3 PMC campaign over the driver's bench command (short form): HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate
# passes) and MFMA-busy / effective clock, per kernel of the ROUND-3 step (attn_bwd_dkdv4, glu_xa, the compile-time-epilogue
# GEMM instances) -> profiles/pmc_traffic.json, profiles/r03z_pmc_*.txt, profiles/r03z_pmc_tables.md
# usage: gpurun --timeout 1200 -- 'bash tools/gpu_r03_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
SECONDS=0
cd /tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --alt-steps 0 --no-cpu-baseline"
pass() {   # tag, counters...
  local tag=$1; shift
  timeout 420 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$tag -o pmc -- $BENCH > $OUT/pmc_$tag.log 2>&1
  echo "pass $tag rc=$? ($SECONDS s)"; tail -1 $OUT/pmc_$tag.log | cut -c1-160
  local DB=$(find $OUT/pmc_$tag -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB > $OUT/r03z_pmc_bench_$tag.txt 2> $OUT/pmc_$tag.err
  rm -rf $OUT/pmc_$tag
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
cd $R
python tools/pmc_to_json.py $OUT/r03z_pmc_bench_fetch.txt $OUT/r03z_pmc_bench_write.txt > $OUT/pmc_traffic.json 2> $OUT/pmc_json.err
{
  echo "# Round-3 PMC tables (in-step, rocprofv3 PMC passes over \`bench.py --steps 2 --warmup 1 --alt-steps 0\`, tools/gpu_r03_pmc.sh)"
  echo; echo "## MFMA-busy and effective clock"; echo
  python tools/pmc_tables.py mfma $OUT/r03z_pmc_bench_mfma.txt
  echo; echo "## HBM-side traffic per launch"; echo
  python tools/pmc_tables.py hbm $OUT/r03z_pmc_bench_fetch.txt $OUT/r03z_pmc_bench_write.txt 8192
} > $OUT/r03z_pmc_tables.md 2> $OUT/pmc_tables.err
head -16 $OUT/r03z_pmc_tables.md | cut -c1-170
echo "pmc done ($SECONDS s)"
