#!/bin/bash
# round 2, 11th GPU pass: decode path (GEMV / RoPE+append / split-KV attention / engine with hipGraph), decode bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decode.py -q -m gpu -x > $OUT/pytest_decode_r02k.log 2>&1
echo "rc=$?"; tail -30 $OUT/pytest_decode_r02k.log
timeout 600 python tools/decode_bench.py --layers 32 --context 2048 --new 64 --out $OUT/decode_r02k.jsonl > $OUT/decode_r02k.log 2>&1
echo "rc=$?"; tail -12 $OUT/decode_r02k.log
