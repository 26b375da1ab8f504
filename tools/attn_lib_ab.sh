#!/bin/bash
# A/B of two builds of the library (UNSLOTH_AMD_LIB) on the attention shapes: usage  attn_lib_ab.sh OTHER.so [knobs] [fwd,bwd]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in $1 libunsloth_amd.so; do
  echo "== $v"
  UNSLOTH_AMD_LIB=$R/unsloth_amd/lib/$v python tools/attn_ab.py ${2:-0} ${3:-bwd} 2>/dev/null | grep '"shape"' | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('  ', r['shape'][:34].ljust(34), r.get('fwd_ms'), r.get('fwd_frac'), r.get('bwd_ms'), r.get('bwd_frac'))"
done
