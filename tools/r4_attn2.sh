#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for th in 0 1 2 3 4 6; do VIPANT_ATTN_DMA_THROTTLE=$th timeout 600 python tools/mha_check.py stag$th 2>&1 | grep "audio\|ViT-L"; done
