#!/bin/bash
# timing experiments on the forward kernel: the shipped library vs debug builds (no K/V DMA; every tile = tile 0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in libunsloth_amd dbg_nodma dbg_sametile; do
  echo "== $v"
  UNSLOTH_AMD_LIB=$R/unsloth_amd/lib/$v.so python tools/attn_ab.py ${1:-16} fwd 2>/dev/null | grep '"shape"' | python3 -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin]
d={}
for r in rows:
    print('  ', r['shape'][:30].ljust(30), r['arm'], r['fwd_ms'], r['fwd_frac'])
    d[(r['shape'][:6],r['arm'])]=r['fwd_ms']
for arm in sorted({k[1] for k in d}):
    a=d[('4x2048',arm)]; b=d[('2x4096',arm)]
    slope=(b-a)*1e-3*2.2e9/64; inter=(a*1e-3*2.2e9-66*slope)/4
    print('   arm',arm,': cycles per tile step at 2.2 GHz',round(slope),'| fixed per block',round(inter))
"
done
