#!/bin/bash
# PMC comparison of the 8-wave and the 4-wave 256-tile GEMM on one shape. usage: gpurun -- 'bash tools/gemm_w4_pmc.sh "M N K"'
SHAPE=${1:-"8192 4096 14336"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for W4 in 0 1; do
 i=0
 for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  UAMD_GEMM_W4=$W4 UAMD_GEMM_PERSIST=0 timeout 200 rocprofv3 --kernel-trace --pmc $SET -d $OUT/w4pmc_${W4}_$i -o pmc -- python $R/tools/gemm_pmc_probe.py $SHAPE > $OUT/w4pmc_${W4}_$i.log 2>&1
  DB=$(find $OUT/w4pmc_${W4}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB gemm_nt256 && rm -rf $OUT/w4pmc_${W4}_$i
 done
done
