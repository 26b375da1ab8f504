#!/bin/bash
# batch-1 step (1 x 2048 tokens) under the GEMM knobs: which tile height / persistence / raster group the small-M shapes want
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; env "$@" python bench.py --only batch1 --steps 6 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d.get('value'), d.get('ms_per_step'), d.get('gemm_frac_of_mfma_peak'))"; }
run A=0
run UAMD_GEMM_HALF=0
run UAMD_GEMM_HALF=2
run UAMD_GEMM_PERSIST=2
run UAMD_GEMM_PERSIST=0
run UAMD_GEMM_GROUP_M=4
run UAMD_GEMM_GROUP_M=16
run UNSLOTH_AMD_GEMM256_MIN_TILES=96
run A=0
