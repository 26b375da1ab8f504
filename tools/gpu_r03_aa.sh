#!/bin/bash
# round 3: nf4_dequant_x4_kernel parity + A/B (microbench, whole step), the reference's self-test grid and the
# two-fresh-process determinism run, bench with the roofline event pairs on every step vs every 4th
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_nf4_gemm.py -m gpu -q -k "nf4_dequantize or bnb_compat" > $OUT/pytest_r03aa_nf4.log 2>&1
echo "pytest nf4 rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_r03aa_nf4.log
timeout 900 python -m pytest tests/test_gpu_reference_selftests.py -m gpu -q -x > $OUT/pytest_r03aa_self.log 2>&1
echo "pytest selftests rc=$? ($SECONDS s)"; tail -25 $OUT/pytest_r03aa_self.log
timeout 300 python tools/dequant_ab.py > $OUT/r03aa_dequant_ab.jsonl 2> $OUT/dequant_ab.err
echo "dequant_ab rc=$? ($SECONDS s)"; cat $OUT/r03aa_dequant_ab.jsonl; tail -3 $OUT/dequant_ab.err
B="python bench.py --gpus 1 --steps 12 --warmup 4 --alt-steps 0 --no-cpu-baseline"
for cfg in "0 1" "1 1" "1 4" "0 4"; do
  set -- $cfg
  UAMD_DEQUANT_X4=$1 timeout 400 $B --roofline-every $2 > $OUT/bench_r03aa_x4_$1_every_$2.json 2> $OUT/bench_r03aa.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_r03aa_x4_$1_every_$2.json").read().strip().splitlines()[-1])
print("x4=$1 roofline-every=$2:", d["value"], "tok/s", d["ms_per_step"], "ms", "gemm", d["roofline"]["achieved"], d["roofline"]["frac"], "launches/step", d["roofline"]["launches_per_step"], "share", d["roofline"]["share_of_step"])
PY
done
echo "all done ($SECONDS s)"
