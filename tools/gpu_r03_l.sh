#!/bin/bash
# GEMM speed: this round's library vs the round-2 gemm256.hip on the SAME box (hipBLASLt as the box-speed reference)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python tools/gemm_ab.py $OUT/gemm_ab_r03l_current.jsonl > /dev/null 2>$OUT/gemm_ab_r03l.err
cat $OUT/gemm_ab_r03l_current.jsonl | head -40
# swap in the round-2 kernel file
cp unsloth_amd/lib/libunsloth_amd.so /tmp/lib_current.so
T=gpurun_tmp_r02
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -ffp-contract=off -I$T/include -I$T/csrc"
/opt/rocm/bin/hipcc $FL -c -x hip $T/csrc/gemm256.hip -o /tmp/gemm256_r02.o && /opt/rocm/bin/hipcc $FL -c -x hip $T/csrc/stub.hip -o /tmp/stub.o
OBJS=$(ls unsloth_amd/lib/*.o | grep -v gemm256.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unsloth_amd/lib/libunsloth_amd.so $OBJS /tmp/gemm256_r02.o /tmp/stub.o
touch unsloth_amd/lib/libunsloth_amd.so
UNSLOTH_AMD_NO_BUILD=1 timeout 300 python tools/gemm_ab.py $OUT/gemm_ab_r03l_r02kernel.jsonl > /dev/null 2>>$OUT/gemm_ab_r03l.err
cat $OUT/gemm_ab_r03l_r02kernel.jsonl | head -40
cp /tmp/lib_current.so unsloth_amd/lib/libunsloth_amd.so
tail -3 $OUT/gemm_ab_r03l.err
