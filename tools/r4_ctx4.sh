#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for d in 0 1 2 3 4 6 7; do echo "dbg=$d"; VIPANT_CTX_DBG=$d timeout 300 python tools/rows_bench.py 2>&1 | grep ctx; done
