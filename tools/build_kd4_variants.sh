#!/bin/bash
# Variant builds of the library for the dK/dV step loop's timing experiments: each regenerates attn_kd4_loop.inc with other
# generator settings and links unsloth_amd/lib/libunsloth_amd_kd4_<tag>.so (A/B through UNSLOTH_AMD_LIB). Restores the shipped loop.
#   tools/build_kd4_variants.sh tag:ENV=VAL,ENV=VAL ...      e.g.  vm58:KD4_VMGAP=58 nodma:KD4_DROP=dma
R=$(cd $(dirname $0)/.. && pwd); cd $R
INC=unsloth_amd/csrc/attn_kd4_loop.inc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -ffp-contract=off -Iinclude -Iunsloth_amd/csrc -fno-slp-vectorize"
OBJS=$(ls unsloth_amd/lib/*.o | grep -v attention.o | grep -v kd4_)
for spec in "$@"; do
  tag=${spec%%:*}; envs=$(echo "${spec#*:}" | tr ';' ' ')
  env $envs python tools/gen/gen_attn_kd4.py > $INC || exit 1
  /opt/rocm/bin/hipcc $FLAGS -c unsloth_amd/csrc/attention.hip -o unsloth_amd/lib/kd4_$tag.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS unsloth_amd/lib/kd4_$tag.o -o unsloth_amd/lib/libunsloth_amd_kd4_$tag.so
  rm unsloth_amd/lib/kd4_$tag.o
  echo "built $tag ($envs): $(sed -n 3p $INC | cut -c1-200)"
done
python tools/gen/gen_attn_kd4.py > $INC
