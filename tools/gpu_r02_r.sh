#!/bin/bash
# round 2, pass r: FlatAdamW tests + A/B, last-step kernel sequence (gaps, short kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_model.py tests/test_gpu_dp_rccl.py -m gpu -q > $OUT/pytest_r02r.log 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest_r02r.log
for flat in 1 0; do
  UNSLOTH_AMD_FLAT_ADAMW=$flat timeout 300 python bench.py --steps 6 --warmup 2 --alt-steps 0 --no-cpu-baseline > $OUT/bench_r02r_flat$flat.json 2> $OUT/bench_r02r_flat$flat.err
  echo "bench flat=$flat rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/bench_r02r_flat$flat.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","peak_vram_gb")}, d["roofline"]["achieved"], d["config"]["optimizer"])
PY
done
UNSLOTH_AMD_FLAT_ADAMW=1 timeout 300 python bench.py --batch 1 --steps 6 --warmup 2 --alt-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch1 flat', d['value'], d['ms_per_step'])"
UNSLOTH_AMD_FLAT_ADAMW=0 timeout 300 python bench.py --batch 1 --steps 6 --warmup 2 --alt-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch1 torch', d['value'], d['ms_per_step'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_r02r -o bench -- python $R/bench.py --steps 2 --warmup 1 --alt-steps 0 --no-cpu-baseline > $OUT/prof_r02r.log 2>&1
cd $R
DB=$(find $OUT/prof_r02r -name '*.db' | head -1)
python tools/rocpd_sequence.py $DB > $OUT/r02r_step_sequence.csv 2> $OUT/r02r_step_sequence.err
tail -40 $OUT/r02r_step_sequence.csv | grep '^#' | cut -c1-220
rm -rf $OUT/prof_r02r
