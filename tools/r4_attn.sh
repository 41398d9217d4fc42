#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python tools/mha_check.py wide > gpurun_out/r4b_wide.txt 2>&1
grep -v Warn gpurun_out/r4b_wide.txt | grep "S=3\|audio\|ViT-L"
VIPANT_ATTN_FWD=16 timeout 600 python tools/mha_check.py old 2>&1 | grep "audio\|ViT-L"
VIPANT_HIP_LIB=$GRAFT_REPO_ROOT/vipant_amd/lib/libvipant_hip_stamps.so python tools/attnw_stamps.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_kernels_gpu.py -q -x -k "mha" 2>&1 | tail -3
