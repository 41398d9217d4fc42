#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "heads or rows_ctx" 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "last_block or golden or trajectory" 2>&1 | tail -4
bash tools/r4_ctx2.sh 2>&1 | grep -E "us  grid" | sed -n 9,22p | cut -c1-140
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-last-block-check 2>/dev/null | head -c 220
