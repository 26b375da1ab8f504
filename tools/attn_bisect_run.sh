#!/bin/bash
# rocprofv3 kernel stats of tools/attn_bwd_prof.py for every unsloth_amd/lib/libbis_*.so + the current library (one lease)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for so in $(ls unsloth_amd/lib/libbis_*.so) unsloth_amd/lib/libunsloth_amd.so; do
  t=$(basename $so .so)
  UNSLOTH_AMD_LIB=$R/$so bash tools/gpu.sh bis_$t pyprof:tools/attn_bwd_prof.py 2>&1 | grep attn | cut -c1-90 | sed "s/^/$t /"
done
