#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "rows_ctx or head_expand or mha_rows" 2>&1 | tail -15 > gpurun_out/r4f_ctx_tests.log
cat gpurun_out/r4f_ctx_tests.log
timeout 300 python tools/rows_bench.py 2>&1 | tee gpurun_out/r4f_rows_bench.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -k "last_block or golden or trajectory" 2>&1 | tail -15 | tee gpurun_out/r4f_model_tests.log
for v in 1 0; do
  VIPANT_LAST_BLOCK_CTX=$v python bench.py --steps 20 --warmup 3 2> gpurun_out/r4f_bench_$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ctx=$v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4f_ab.txt
done
