#!/bin/bash
# An alternate build of ONE source file into its own library, for A/B timing through VIPANT_HIP_LIB (the default library is untouched):
#   bash tools/build_variant.sh attention.hip -DVIPANT_ATTN_FWD_KV_SPLIT kvsplit   -> vipant_amd/lib/libvipant_hip_kvsplit.so
set -e
cd "$(dirname "$0")/.."
src=$1; def=$2; tag=$3
python -m vipant_amd.build > /dev/null
extra=""
case "$src" in
  attention.hip) extra="-mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm" ;;
  gemm_nt.hip) extra="-mllvm -disable-machine-sink -mllvm -amdgpu-atomic-optimizer-strategy=None" ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $extra $def -c vipant_amd/csrc/$src -o /tmp/variant_$tag.o
objs=$(ls vipant_amd/lib/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vipant_amd/lib/libvipant_hip_$tag.so $objs /tmp/variant_$tag.o
echo vipant_amd/lib/libvipant_hip_$tag.so
