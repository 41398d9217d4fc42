#!/bin/bash
# debug build of the library with in-kernel cycle stamps in the attention kernels: vipant_amd/lib/libvipant_hip_stamps.so
set -e
cd "$(dirname "$0")/.."
python -m vipant_amd.build > /dev/null
O=vipant_amd/lib/obj
for f in attention attention_wide; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DVIPANT_ATTN_STAMPS -c vipant_amd/csrc/$f.hip -o /tmp/${f}_stamps.o &
done
wait
OBJS=$(ls $O/*.o | grep -v "/attention.o" | grep -v "/attention_wide.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vipant_amd/lib/libvipant_hip_stamps.so $OBJS /tmp/attention_stamps.o /tmp/attention_wide_stamps.o
echo vipant_amd/lib/libvipant_hip_stamps.so
