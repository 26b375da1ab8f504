#!/bin/bash
# decode step, two builds of the library on one box, alternating: usage decode_lib_ab.sh OTHER.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2 3; do
  for v in $1 libunsloth_amd.so; do
    echo -n "$v: "
    UNSLOTH_AMD_LIB=$R/unsloth_amd/lib/$v python tools/decode_bench.py --quick 2>/dev/null | grep "decode tokens" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_token'], d['tokens_per_s'])"
  done
done
