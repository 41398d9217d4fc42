#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VIPANT_ATTN_FWD=16
VIPANT_ATTN_BWD=4 timeout 300 python tools/mha_check.py wide 2>&1 | grep -v Warn | grep "S=3\|audio\|ViT-L"
VIPANT_ATTN_BWD=3 timeout 300 python tools/mha_check.py stream 2>&1 | grep "audio\|ViT-L"
