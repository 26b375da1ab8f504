#!/bin/bash
# attention library A/B in one lease: attn_ab_run.sh TAG  ->  gpurun_out/TAG.jsonl (prev / lib interleaved, backward then forward)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-attn_ab}
for rep in 1 2; do
  for what in bwd fwd; do
    for v in libunsloth_amd_prev.so libunsloth_amd.so; do
      UNSLOTH_AMD_LIB=$R/unsloth_amd/lib/$v python tools/attn_bwd_time.py ${v%.so} $what 2>/dev/null | tee -a $OUT/$TAG.jsonl
    done
  done
done
