#!/bin/bash
# PMC passes over tools/gemm_pmc_probe.py (ours 8-wave, ours 4-wave, hipBLASLt on one shape) -> gpurun_out/TAG_gemm_pmc_*.txt
#   gpurun -- 'bash tools/gpu_gemm_pmc.sh TAG [M N K]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
TAG=$1; shift
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]*LDS[A-Z_0-9]*|SQ_INST[A-Z_0-9]*|SQ_ACTIVE_INST[A-Z_0-9]*|TCC_[A-Z_0-9]*(HIT|MISS|REQ)[A-Z_0-9]*|TCP_[A-Z_0-9]*)\b" | sort -u > $OUT/${TAG}_counter_names.txt
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
            "FETCH_SIZE TCC_HIT_sum" "TCC_MISS_sum TCC_REQ_sum" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT/pmc_g$i -o pmc -- python $R/tools/gemm_pmc_probe.py "$@" > $OUT/pmc_${TAG}_g$i.log 2>&1
  echo "[pass $i: $pass] rc=$?"
  DB=$(find $OUT/pmc_g$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB 2>&1 | grep -A 12 -E "gemm_nt256|Cijk" > $OUT/${TAG}_gemm_pmc_pass$i.txt
  rm -rf $OUT/pmc_g$i
  tail -3 $OUT/pmc_${TAG}_g$i.log | cut -c1-200
done
cat $OUT/${TAG}_gemm_pmc_pass*.txt | cut -c1-150
