#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 0 1048576 1 0 1048576; do echo "variant $v"; VIPANT_GEMM_VARIANT=$v timeout 300 python tools/lib_ab.py 2>&1 | grep c_fc; done
