#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for th in 0 97 96; do VIPANT_ATTN_DMA_THROTTLE=$th timeout 600 python tools/mha_check.py probe$th 2>&1 | grep "audio\|ViT-L"; done
