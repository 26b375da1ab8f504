#!/bin/bash
# round 3: model-level regression at HEAD after the PLAIN persistent-GEMM instance and the auto policy (the whole suite ran
# green two commits of kernels earlier: tools/gpu_r03_final2.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_lora_blocks.py tests/test_gpu_baseline_configs.py tests/test_gpu_nf4_gemm.py tests/test_gpu_glu_fused.py tests/test_gpu_optim.py tests/test_gpu_api_surface.py -m gpu -q -x > $OUT/pytest_r03af.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_r03af.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "all done ($SECONDS s)"
