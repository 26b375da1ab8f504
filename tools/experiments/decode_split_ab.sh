#!/bin/bash
# decode step under different split sizes of the fused attention, one box, alternating
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for k in 128 64 32; do
    echo -n "split_keys=$k: "
    UNSLOTH_AMD_DECODE_SPLIT_KEYS=$k python tools/decode_bench.py --quick 2>/dev/null | grep "decode tokens" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_token'], d['tokens_per_s'])"
  done
done
