#!/bin/bash
# PMC passes for the GEMM probe. usage: gpurun -- 'bash tools/gpu_pmc.sh tag "M N K"'
TAG=${1:-pmc}; SHAPE=${2:-"8192 14336 4096"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" ${PMC_EXTRA}; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/${TAG}_$i -o pmc -- python $R/tools/gemm_pmc_probe.py $SHAPE > $OUT/${TAG}_$i.log 2>&1
  tail -1 $OUT/${TAG}_$i.log
  DB=$(find $OUT/${TAG}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB gemm > $OUT/${TAG}_$i.txt 2>&1 && cat $OUT/${TAG}_$i.txt
  [ -n "$DB" ] && rm -f $DB
done
