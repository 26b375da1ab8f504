#!/bin/bash
# round 2, first GPU pass: bf16 golden from the reference's Triton kernels (native), parity vs it, the reference's
# kernel timings, hipBLASLt kernel names on the step's shapes, full -m gpu regression.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python oracle/make_golden_bf16_gpu.py --out $OUT/ref_triton_bf16.pt > $OUT/golden_bf16.log 2>&1
tail -15 $OUT/golden_bf16.log
cp $OUT/ref_triton_bf16.pt tests/golden/ 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_ref_bf16_golden.py -q -m gpu > $OUT/pytest_bf16_golden.log 2>&1
tail -30 $OUT/pytest_bf16_golden.log
timeout 300 python oracle/bench_reference_triton.py --out $OUT/ref_triton_bench.jsonl > $OUT/ref_triton_bench.log 2>&1
tail -12 $OUT/ref_triton_bench.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_hipblaslt -o probe -- python $R/tools/hipblaslt_probe.py > $OUT/hipblaslt_probe.log 2>&1 )
tail -8 $OUT/hipblaslt_probe.log
DB=$(find $OUT/prof_hipblaslt -name '*.db' | head -1)
python tools/rocpd_names.py $DB > $OUT/hipblaslt_kernels.txt 2>&1
head -30 $OUT/hipblaslt_kernels.txt
find $OUT/prof_hipblaslt -size +8M -delete
timeout 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_ref_bf16_golden.py > $OUT/pytest_gpu_r02a.log 2>&1
tail -3 $OUT/pytest_gpu_r02a.log
