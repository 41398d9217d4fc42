#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "rows_ctx or head_expand" 2>&1 | tail -12
timeout 300 python tools/rows_bench.py 2>&1 | grep ctx
