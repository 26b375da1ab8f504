#!/bin/bash
# NEXT STEP named in DESIGN 10 (not run yet: the round's GPU budget was spent): MFMA-busy and effective clock of hipBLASLt's
# stream-K kernel beside ours on the step's GEMM shapes, same process, so that the 5-8 % it has on this round's boxes can be
# attributed (higher clock at the same MFMA-busy = less power per flop; higher MFMA-busy at the same clock = denser issue).
# usage: gpurun --timeout 600 -- 'bash tools/gpu_next_gemm_yardstick.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout ${YARDSTICK_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    -d $OUT/pmc_gemm_yardstick -o pmc -- python $R/tools/gemm_plain_ab.py > $OUT/gemm_yardstick.log 2>&1
tail -8 $OUT/gemm_yardstick.log | cut -c1-220
cd $R
DB=$(find $OUT/pmc_gemm_yardstick -name '*.db' | head -1)
[ -n "$DB" ] && python tools/pmc_summary.py $DB > $OUT/gemm_yardstick_pmc.txt 2> $OUT/gemm_yardstick_pmc.err && rm -rf $OUT/pmc_gemm_yardstick
python tools/pmc_tables.py mfma $OUT/gemm_yardstick_pmc.txt | head -12
# (the rows to compare: Custom_Cijk_Alik_Bljk_..._MT256x256x64 = hipBLASLt; gemm_nt256p_kernel / gemm_nt256_kernel = ours)
