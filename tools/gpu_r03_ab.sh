#!/bin/bash
# round 3: glu_xa_kernel with the branch-free tile loop (no vmcnt(0) at the loop header), CE forward with two loads in
# flight, CE backward with a branch-free full chunk: parity tests, microbench, in-step kernel stats, whole-step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_glu_fused.py tests/test_gpu_elementwise.py tests/test_gpu_ref_bf16_golden.py tests/test_gpu_rl_logprobs.py -m gpu -q -x > $OUT/pytest_r03ab.log 2>&1
echo "pytest rc=$? ($SECONDS s)"; tail -5 $OUT/pytest_r03ab.log
timeout 300 python tools/glu_fused_bench.py > $OUT/r03ab_glu_fused_bench.jsonl 2> $OUT/glu_bench.err
echo "glu bench rc=$? ($SECONDS s)"; cat $OUT/r03ab_glu_fused_bench.jsonl; tail -2 $OUT/glu_bench.err
B="python bench.py --gpus 1 --steps 12 --warmup 4 --alt-steps 0 --no-cpu-baseline"
for mode in bwd both bwd both; do
  UNSLOTH_AMD_GLU_FUSED=$mode timeout 400 $B > $OUT/bench_r03ab_$mode.json 2> $OUT/bench_r03ab.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_r03ab_$mode.json").read().strip().splitlines()[-1])
print("GLU_FUSED=$mode:", d["value"], "tok/s", d["ms_per_step"], "ms", "gemm", d["roofline"]["achieved"], d["roofline"]["frac"])
PY
done
cd /tmp
UNSLOTH_AMD_GLU_FUSED=both timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r03ab -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 > $OUT/prof_r03ab.log 2>&1
cd $R
DB=$(find $OUT/prof_r03ab -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > $OUT/r03ab_bench_kernel_stats.csv 2> $OUT/r03ab_stats.err
python tools/rocpd_sequence.py $DB > $OUT/r03ab_step_sequence.csv 2> $OUT/r03ab_seq.err
grep "glu_\|ce_\|nf4_dequant\|lora_xa2" $OUT/r03ab_bench_kernel_stats.csv | cut -c1-120; grep "^# kernels" $OUT/r03ab_step_sequence.csv
rm -rf $OUT/prof_r03ab
echo "all done ($SECONDS s)"
