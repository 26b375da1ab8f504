#!/bin/bash
# lora_tn library A/B in one lease: lora_tn_ab_run.sh TAG lib1.so lib2.so ...  ->  gpurun_out/TAG.jsonl (interleaved, twice)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
TAG=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    UNSLOTH_AMD_LIB=$R/unsloth_amd/lib/$v python tools/lora_tn_ab.py ${v%.so} 2>/dev/null | tee -a $OUT/$TAG.jsonl
  done
done
