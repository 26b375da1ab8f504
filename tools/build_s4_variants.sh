#!/bin/bash
# Builds one library per K-loop schedule of gemm_nt256s_kernel (tools/gen/gen_gemm256s.py SCHEDULES) for an A/B on one box:
#   bash tools/build_s4_variants.sh vendor even3 ...   ->  unsloth_amd/lib/libunsloth_amd_s4_<name>.so  (knock-outs compiled in)
# The LAST name given stays as the in-tree gemm256s_loop.inc / libunsloth_amd.so.
set -e
cd "$(dirname "$0")/.."
for s in "$@"; do
  G256S_SCHED=$s python tools/gen/gen_gemm256s.py > unsloth_amd/csrc/gemm256s_loop.inc
  UAMD_EXTRA_CFLAGS=-DUAMD_G256S_KNOCKOUTS python -c "from unsloth_amd import _build; _build.build()" > /dev/null
  cp unsloth_amd/lib/libunsloth_amd.so unsloth_amd/lib/libunsloth_amd_s4_$s.so
  echo built $s
done
