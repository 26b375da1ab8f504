#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_glu_fused.py tests/test_gpu_lora_blocks.py tests/test_gpu_model.py -m gpu -q -x --durations=5 > $OUT/pytest_r03h.log 2>&1
echo "pytest rc=$?"; tail -30 $OUT/pytest_r03h.log
# order-dependence of the config-5 log-prob gradients (1.41 in the full suite, 0.021 alone)
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_baseline_fullsize.py -m gpu -q -k "attention or config5" > $OUT/pytest_r03h2.log 2>&1
echo "attention+config5 rc=$?"; tail -4 $OUT/pytest_r03h2.log; python -c "
import json; d=json.load(open('gpurun_out/fullsize_parity.json')); print('total grad', d['config5_logprobs']['total_grad_rel_fro'])"
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_baseline_fullsize.py -m gpu -q -k "widths or config5" > $OUT/pytest_r03h3.log 2>&1
echo "configs+config5 rc=$?"; tail -4 $OUT/pytest_r03h3.log; python -c "
import json; d=json.load(open('gpurun_out/fullsize_parity.json')); print('total grad', d['config5_logprobs']['total_grad_rel_fro'])"
# A/B of the fusion in the step
for F in 1 0; do
  UNSLOTH_AMD_GLU_FUSED=$F timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --alt-steps 0 > $OUT/bench_r03h_fused$F.json 2> $OUT/bench_r03h_fused$F.err
  python -c "
import json; d=json.loads(open('$OUT/bench_r03h_fused$F.json').read().strip().splitlines()[-1]); print('fused=$F', d['value'], d['ms_per_step'])"
done
