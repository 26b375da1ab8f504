#!/bin/bash
# PMC passes over tools/attn_bench.py (attention kernels). usage: gpurun -- 'bash tools/gpu_pmc_attn.sh tag'
TAG=${1:-pmc_attn}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/${TAG}_$i -o pmc -- python $R/tools/attn_bench.py > $OUT/${TAG}_$i.log 2>&1
  DB=$(find $OUT/${TAG}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB attn_ > $OUT/${TAG}_$i.txt 2>&1 && cat $OUT/${TAG}_$i.txt
  [ -n "$DB" ] && rm -f $DB
done
