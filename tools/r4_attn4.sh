#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VIPANT_ATTN_FWD=16
for v in 7 8 9; do VIPANT_ATTN_STAGGER=$v timeout 300 python tools/mha_check.py var$v 2>&1 | grep -v Warn | grep "S=316\|audio"; done
